// Micro-benchmark for the persistent round kernel (frx_round_kernel.hpp): the price of the hand-offs it is built from, measured
// in its own geometry.  B clusters of G workgroups (256 threads, 140 KB of LDS each => one per CU); per round
//     leader (workgroup 0 of the cluster)  [polls a word of mapped host memory]  publishes NX doubles with write-through stores + a flag
//     every workgroup                      polls the flag, reads its chunk back with L1-bypassing loads, checks every word,
//                                          publishes 512 doubles of "partials", arrives on the leader's counter
//     leader                               waits for the G arrivals, reads all G x 512 partials, checks them  [posts to the host]
// All payload traffic uses the {sc1 store, drain, flag} / {relaxed poll, sc1 load} form of the CDNA guide (Guideline 16, R1 with
// L1-bypassing loads instead of the acquire).  Every spin is bounded.  Prints us per round (device only, and with the host in the loop).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM

__device__ __forceinline__ void st_wt(double *p, double v) { __hip_atomic_store((u64 *)p, (u64)__double_as_longlong(v), RLX_AGENT); }
__device__ __forceinline__ double ld_l2(const double *p) { return __longlong_as_double((long long)__hip_atomic_load((const u64 *)p, RLX_AGENT)); }

// one lane polls until *w == want (agent scope) or the deadline passes; returns false on timeout
// (also gives up as soon as anybody else has recorded a timeout in *abort, so one lost hand-off ends the whole launch quickly)
__device__ __forceinline__ bool wait_eq(const unsigned *w, unsigned want, u64 deadline, const unsigned *abort) {
    for (unsigned spins = 0;; spins++) {
        if (__hip_atomic_load(w, RLX_AGENT) == want) return true;
        if ((spins & 63) == 63 && (wall_clock64() > deadline || __hip_atomic_load(abort, RLX_AGENT) != 0)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

struct Probe {
    double *pub;            // [B][NX]        leader -> cluster
    double *part;           // [B][G][512]    cluster -> leader
    unsigned *flag;         // [B]            epoch of pub
    unsigned *arrive;       // [B]            arrivals (monotone)
    unsigned *census;       // [1]
    unsigned *err;          // [4]  0: timeouts, 1: payload mismatches (cluster side), 2: mismatches (leader side), 3: census failures
    volatile unsigned *h_cmd;   // [B] mapped host memory: host -> device round number
    volatile unsigned *h_res;   // [B] mapped host memory: device -> host round number
    u64 *cycles;            // [B] leader: wall-clock ticks (100 MHz) spent in the timed rounds
    int B, G, NX, rounds, with_host;
};

__global__ __launch_bounds__(256) void k_probe(Probe pr) {
    extern __shared__ double sm[];
    const int c = blockIdx.x / pr.G, g = blockIdx.x % pr.G, t = threadIdx.x;
    const u64 t_start = wall_clock64();
    const u64 tmo = 100000000ull * 2;                                        // 2 s at 100 MHz
    __shared__ int ok;
    if (t == 0) {
        atomicAdd(pr.census, 1u);
        ok = wait_eq(pr.census, gridDim.x, t_start + tmo, pr.err) ? 1 : 0;
        if (!ok) { atomicAdd(pr.err + 3, 1u); atomicAdd(pr.err, 1u); }
    }
    __syncthreads();
    if (!ok) return;
    const int CH = (pr.NX + pr.G - 1) / pr.G;
    double *pub = pr.pub + (size_t)c * pr.NX, *part = pr.part + (size_t)c * pr.G * 512;
    u64 t0 = 0;
    for (int r = 1; r <= pr.rounds; r++) {
        if (t == 0) ok = __hip_atomic_load(pr.err, RLX_AGENT) == 0;
        __syncthreads();
        if (!ok) break;
        if (r == 11 && g == 0 && t == 0) t0 = wall_clock64();
        if (g == 0) {
            if (pr.with_host && t == 0) {
                const u64 dl = wall_clock64() + tmo;
                for (unsigned spins = 0;; spins++) {
                    if (__hip_atomic_load((const unsigned *)pr.h_cmd + c, RLX_SYS) == (unsigned)r) break;
                    if ((spins & 63) == 63 && (wall_clock64() > dl || __hip_atomic_load(pr.err, RLX_AGENT) != 0)) { atomicAdd(pr.err, 1u); break; }
                }
            }
            __syncthreads();
            for (int i = t; i < pr.NX; i += 256) st_wt(pub + i, r * 1000.0 + i);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(pr.flag + c, (unsigned)r, RLX_AGENT);
        }
        if (t == 0 && !wait_eq(pr.flag + c, (unsigned)r, wall_clock64() + tmo, pr.err)) atomicAdd(pr.err, 1u);
        __syncthreads();
        {
            int bad = 0;
            for (int i = g * CH + t; i < min((g + 1) * CH, pr.NX); i += 256) bad += ld_l2(pub + i) != r * 1000.0 + i;
            if (bad) atomicAdd(pr.err + 1, (unsigned)bad);
            for (int i = t; i < 512; i += 256) st_wt(part + (size_t)g * 512 + i, r + 0.001 * i + g);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(pr.arrive + c, 1u, RLX_AGENT);
        }
        if (g == 0) {
            if (t == 0 && !wait_eq(pr.arrive + c, (unsigned)(pr.G * r), wall_clock64() + tmo, pr.err)) atomicAdd(pr.err, 1u);
            __syncthreads();
            int bad = 0;
            for (int i = t; i < 512 * pr.G; i += 256) bad += ld_l2(part + i) != r + 0.001 * (i & 511) + (i >> 9);
            if (bad) atomicAdd(pr.err + 2, (unsigned)bad);
            if (pr.with_host && t == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                __hip_atomic_store((unsigned *)pr.h_res + c, (unsigned)r, RLX_SYS);
            }
        }
    }
    if (g == 0 && t == 0) pr.cycles[c] = wall_clock64() - t0;
    if (sm[t] == 123.0) pr.cycles[c] = 0;                                    // keeps the dynamic LDS request alive
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int NX = 704, rounds = 2010;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const size_t lds = 140 * 1024;
    CK(hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int cfg = 0; cfg < 8; cfg++) {
        const int Bs[] = {32, 32, 1, 1, 8, 16, 32, 28}, Gs[] = {8, 8, 8, 8, 8, 8, 4, 8}, hosts[] = {0, 1, 0, 1, 1, 1, 1, 1};
        Probe pr; pr.B = Bs[cfg]; pr.G = Gs[cfg]; pr.NX = NX; pr.rounds = rounds; pr.with_host = hosts[cfg];
        CK(hipMalloc(&pr.pub, sizeof(double) * pr.B * NX)); CK(hipMalloc(&pr.part, sizeof(double) * pr.B * pr.G * 512));
        CK(hipMalloc(&pr.flag, 4 * pr.B)); CK(hipMalloc(&pr.arrive, 4 * pr.B)); CK(hipMalloc(&pr.census, 4)); CK(hipMalloc(&pr.err, 16)); CK(hipMalloc(&pr.cycles, 8 * pr.B));
        CK(hipMemset(pr.flag, 0, 4 * pr.B)); CK(hipMemset(pr.arrive, 0, 4 * pr.B)); CK(hipMemset(pr.census, 0, 4)); CK(hipMemset(pr.err, 0, 16)); CK(hipMemset(pr.cycles, 0, 8 * pr.B));
        unsigned *hc, *hr;
        CK(hipHostMalloc((void **)&hc, 4 * pr.B, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostMalloc((void **)&hr, 4 * pr.B, hipHostMallocMapped | hipHostMallocCoherent));
        for (int b = 0; b < pr.B; b++) { hc[b] = 0; hr[b] = 0; }
        pr.h_cmd = hc; pr.h_res = hr;
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        const auto w0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_probe, dim3(pr.B * pr.G), dim3(256), lds, st, pr);
        CK(hipGetLastError());
        bool host_tmo = false;
        if (pr.with_host) {
            // per-candidate asynchronous mailbox: answer every candidate as soon as its result shows up
            std::vector<unsigned> next(pr.B, 1);
            for (int b = 0; b < pr.B; b++) { std::atomic_thread_fence(std::memory_order_release); hc[b] = 1; }
            int done = 0;
            const auto dl = std::chrono::steady_clock::now() + std::chrono::seconds(8);
            while (done < pr.B && !host_tmo) {
                for (int b = 0; b < pr.B; b++) {
                    if (next[b] > (unsigned)rounds) continue;
                    if (*(volatile unsigned *)(hr + b) == next[b]) { next[b]++; if (next[b] > (unsigned)rounds) done++; else hc[b] = next[b]; }
                }
                if (std::chrono::steady_clock::now() > dl) host_tmo = true;
            }
        }
        CK(hipStreamSynchronize(st));
        const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        unsigned err[4]; std::vector<u64> cyc(pr.B);
        CK(hipMemcpy(err, pr.err, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(cyc.data(), pr.cycles, 8 * pr.B, hipMemcpyDeviceToHost));
        double mean = 0, mx = 0;
        for (int b = 0; b < pr.B; b++) { const double us = cyc[b] / 100.0 / (rounds - 10); mean += us / pr.B; mx = us > mx ? us : mx; }
        printf("B=%2d G=%d host=%d: %.2f us/round mean, %.2f max over clusters (wall %.1f ms)  timeouts %u mismatches %u/%u census_fail %u%s\n", pr.B, pr.G, pr.with_host,
               mean, mx, wall_ms, err[0], err[1], err[2], err[3], host_tmo ? "  HOST TIMEOUT" : "");
        fflush(stdout);
        hipFree(pr.pub); hipFree(pr.part); hipFree(pr.flag); hipFree(pr.arrive); hipFree(pr.census); hipFree(pr.err); hipFree(pr.cycles); hipHostFree(hc); hipHostFree(hr); hipStreamDestroy(st);
    }
    return 0;
}
