// Micro-benchmark for the host mailbox of the resident round kernel: where should the COMMAND word live?
//   mode 0: mapped host memory (hipHostMalloc), the device polls it over PCIe              (what frx_round_kernel.hpp did in round 1)
//   mode 1: fine-grained device memory (hipExtMallocWithFlags), written by the host through the PCIe BAR, polled by the device in its own memory
//   mode 2: plain hipMalloc memory written by the host through the BAR
// B pollers (one workgroup each, one lane polling, bounded), results always go to mapped host memory (posted writes).  The host serves
// all B mailboxes from one thread and waits `think_us` between receiving a result and posting the next command for a quarter of
// the rounds, as the line search does.  Prints the device-side wait per round (ticks of wall_clock64 = 10 ns) and the wall time.
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned long long u64;
#define RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_poll(volatile u64 *cmd, u64 *res, u64 *waited, int rounds, int work_iters, int sleep_between) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    u64 acc = 0;
    const u64 deadline = wall_clock64() + 200000000ull;            // 2 s
    for (int r = 1; r <= rounds; r++) {
        const u64 t0 = wall_clock64();
        for (;;) {
            const u64 w = __hip_atomic_load((const u64 *)&cmd[b * 8], RLX_SYS);
            if (w == (u64)r) break;
            if (wall_clock64() > deadline) { waited[b] = ~0ull; return; }
            if (sleep_between) __builtin_amdgcn_s_sleep(8);
        }
        acc += wall_clock64() - t0;
        const u64 t1 = wall_clock64();
        while (wall_clock64() - t1 < (u64)work_iters) {}             // the round's work: work_iters x 10 ns
        __hip_atomic_store(&res[b * 8], (u64)r, RLX_SYS);
    }
    waited[b] = acc;
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, rounds = argc > 2 ? atoi(argv[2]) : 2000, work = argc > 3 ? atoi(argv[3]) : 3000;
    for (int mode = 0; mode < 3; mode++)
        for (int slp = 0; slp < 2; slp++) {
            u64 *cmd = nullptr, *res = nullptr, *waited = nullptr;
            if (mode == 0) CK(hipHostMalloc((void **)&cmd, B * 64, hipHostMallocMapped));
            else if (mode == 1) { if (hipExtMallocWithFlags((void **)&cmd, B * 64, hipDeviceMallocFinegrained) != hipSuccess) { printf("mode 1: no fine-grained device memory\n"); continue; } }
            else CK(hipMalloc((void **)&cmd, B * 64));
            CK(hipHostMalloc((void **)&res, B * 64, hipHostMallocMapped));
            CK(hipHostMalloc((void **)&waited, B * 8, hipHostMallocMapped));
            CK(hipMemset(cmd, 0, B * 64));
            memset(res, 0, B * 64); memset(waited, 0, B * 8);
            signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
            if (sigsetjmp(jb, 1)) { printf("mode %d: the host cannot write this memory (signal)\n", mode); signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL); break; }
            volatile u64 *hc = cmd, *hr = res;
            hc[0] = 0;                                                 // first host access: faults here if the memory is not host-visible
            hipLaunchKernelGGL(k_poll, dim3(B), dim3(64), 0, 0, cmd, res, waited, rounds, work, slp);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<int> next(B, 1);
            int done = 0;
            for (int b = 0; b < B; b++) hc[b * 8] = 1;
            while (done < B) {
                for (int b = 0; b < B; b++) {
                    if (next[b] > rounds) continue;
                    if (hr[b * 8] == (u64)next[b]) {
                        next[b]++;
                        if (next[b] > rounds) { done++; continue; }
                        hc[b * 8] = (u64)next[b];
                    }
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) { printf("host timeout\n"); break; }
            }
            CK(hipDeviceSynchronize());
            const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            double w = 0; for (int b = 0; b < B; b++) w += (double)waited[b];
            printf("mode %d (%s) sleep %d  B %d: device waits %.2f us per round for its command, wall %.2f us per round (work %.1f us)\n", mode,
                   mode == 0 ? "mapped host memory" : mode == 1 ? "fine-grained device memory" : "hipMalloc", slp, B, w / B / rounds * 0.01, wall / rounds, work * 0.01);
            signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL);
            if (mode == 0) hipHostFree(cmd); else hipFree(cmd);
            hipHostFree(res); hipHostFree(waited);
        }
    return 0;
}
