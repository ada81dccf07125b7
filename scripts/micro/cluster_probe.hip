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

// R2 form: a double travels as two 8-byte granules {epoch : 32 | half of the bits : 32}; the consumer re-reads until both tags match.
__device__ __forceinline__ void st_gran(u64 *p, double v, unsigned epoch) {
    const u64 b = (u64)__double_as_longlong(v), e = (u64)epoch << 32;
    __hip_atomic_store(p, e | (b & 0xffffffffull), RLX_AGENT);
    __hip_atomic_store(p + 1, e | (b >> 32), RLX_AGENT);
}
__device__ __forceinline__ bool ld_gran(const u64 *p, unsigned epoch, double &v, u64 deadline, unsigned *abort, unsigned site = 0) {
    for (unsigned spins = 0;; spins++) {
        const u64 a = __hip_atomic_load(p, RLX_AGENT), b = __hip_atomic_load(p + 1, RLX_AGENT);
        if ((unsigned)(a >> 32) == epoch && (unsigned)(b >> 32) == epoch) { v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32))); return true; }
        if ((spins & 63) == 63 && (wall_clock64() > deadline || __hip_atomic_load(abort, RLX_AGENT) != 0)) {
            if (atomicCAS(abort + 4, 0u, site | 0x80000000u) == 0u) { abort[5] = (unsigned)(a >> 32); abort[6] = (unsigned)(b >> 32); abort[7] = epoch; }
            return false;
        }
    }
}
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
    int B, G, NX, rounds, with_host, granules, xcd_map;       // xcd_map: cluster c = blocks (c % 8) + 8 (g + G (c / 8)); plain stores when the census says one XCD
    unsigned *xcc;          // [B][G] XCC id of every workgroup
    u64 *gpub, *gpart;      // granule buffers: [B][2 NX], [B][G][2 * 512]
};

__global__ __launch_bounds__(256) void k_probe(Probe pr) {
    extern __shared__ double sm[];
    int c = blockIdx.x / pr.G, g = blockIdx.x % pr.G;
    const int t = threadIdx.x;
    if (pr.xcd_map) { const int lane8 = blockIdx.x & 7, rest = blockIdx.x >> 3; g = rest % pr.G; c = lane8 + 8 * (rest / pr.G); }
    unsigned my_xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 15u;
    if (t == 0) { __hip_atomic_store(pr.xcc + c * pr.G + g, my_xcc + 1u, RLX_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
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
    // after the census every workgroup's XCC id is published: same XCD for the whole cluster => plain (L2-coherent) payload stores
    __shared__ int same_xcd;
    if (t == 0) { int same = pr.xcd_map; for (int k = 0; k < pr.G; k++) same &= (__hip_atomic_load(pr.xcc + c * pr.G + k, RLX_AGENT) == my_xcc + 1u); same_xcd = same; if (same && g == 0) atomicAdd(pr.err + 8, 1u); }
    __syncthreads();
    const bool fast = same_xcd != 0;
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
            if (pr.granules) { for (int i = t; i < pr.NX; i += 256) st_gran(pr.gpub + ((size_t)c * pr.NX + i) * 2, r * 1000.0 + i, (unsigned)r); }
            else {
            if (fast) { for (int i = t; i < pr.NX; i += 256) pub[i] = r * 1000.0 + i; } else
            for (int i = t; i < pr.NX; i += 256) st_wt(pub + i, r * 1000.0 + i);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(pr.flag + c, (unsigned)r, RLX_AGENT);
            }
        }
        if (pr.granules) {
            int bad = 0;
            const u64 dl = wall_clock64() + tmo;
            for (int i = g * CH + t; i < min((g + 1) * CH, pr.NX); i += 256) { double v = 0.0; if (!ld_gran(pr.gpub + ((size_t)c * pr.NX + i) * 2, (unsigned)r, v, dl, pr.err, (1u << 24) | (g << 16) | i)) atomicAdd(pr.err, 1u); bad += v != r * 1000.0 + i; }
            if (bad) atomicAdd(pr.err + 1, (unsigned)bad);
            __syncthreads();                                                     // the partials of round r only after this workgroup has consumed pub[r]
            for (int i = t; i < 512; i += 256) st_gran(pr.gpart + (((size_t)c * pr.G + g) * 512 + i) * 2, r + 0.001 * i + g, (unsigned)r);
            if (g == 0) {
                bad = 0;
                for (int i = t; i < 512 * pr.G; i += 256) { double v = 0.0; if (!ld_gran(pr.gpart + ((size_t)c * pr.G * 512 + i) * 2, (unsigned)r, v, dl, pr.err, (2u << 24) | i)) atomicAdd(pr.err, 1u); bad += v != r + 0.001 * (i & 511) + (i >> 9); }
                if (bad) atomicAdd(pr.err + 2, (unsigned)bad);
                __syncthreads();
                if (pr.with_host && t == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); __hip_atomic_store((unsigned *)pr.h_res + c, (unsigned)r, RLX_SYS); }
            }
            __syncthreads();
            continue;
        }
        if (t == 0 && !wait_eq(pr.flag + c, (unsigned)r, wall_clock64() + tmo, pr.err)) atomicAdd(pr.err, 1u);
        __syncthreads();
        {
            int bad = 0;
            for (int i = g * CH + t; i < min((g + 1) * CH, pr.NX); i += 256) bad += ld_l2(pub + i) != r * 1000.0 + i;
            if (bad) atomicAdd(pr.err + 1, (unsigned)bad);
            if (fast) { for (int i = t; i < 512; i += 256) part[(size_t)g * 512 + i] = r + 0.001 * i + g; } else
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
    int ncfg = 8;
    int Bs[] = {32, 32, 32, 32, 8, 8, 16, 32}, Gs[] = {8, 8, 8, 8, 8, 8, 8, 7}, hosts[] = {0, 0, 1, 1, 0, 0, 0, 0}, grans[] = {0, 0, 0, 0, 0, 0, 0, 0}, xmaps[] = {0, 1, 0, 1, 0, 1, 1, 1};
    if (argc >= 5) { ncfg = 1; Bs[0] = atoi(argv[1]); Gs[0] = atoi(argv[2]); hosts[0] = atoi(argv[3]); grans[0] = atoi(argv[4]); }
    for (int cfg = 0; cfg < ncfg; cfg++) {
        Probe pr; pr.B = Bs[cfg]; pr.G = Gs[cfg]; pr.NX = NX; pr.rounds = rounds; pr.with_host = hosts[cfg]; pr.granules = grans[cfg]; pr.xcd_map = xmaps[cfg];
        CK(hipMalloc(&pr.xcc, 4 * pr.B * pr.G)); CK(hipMemset(pr.xcc, 0, 4 * pr.B * pr.G));
        CK(hipMalloc(&pr.gpub, 16 * (size_t)pr.B * NX)); CK(hipMalloc(&pr.gpart, 16 * (size_t)pr.B * pr.G * 512)); CK(hipMemset(pr.gpub, 0, 16 * (size_t)pr.B * NX)); CK(hipMemset(pr.gpart, 0, 16 * (size_t)pr.B * pr.G * 512));
        CK(hipMalloc(&pr.pub, sizeof(double) * pr.B * NX)); CK(hipMalloc(&pr.part, sizeof(double) * pr.B * pr.G * 512));
        CK(hipMalloc(&pr.flag, 4 * pr.B)); CK(hipMalloc(&pr.arrive, 4 * pr.B)); CK(hipMalloc(&pr.census, 4)); CK(hipMalloc(&pr.err, 64)); CK(hipMalloc(&pr.cycles, 8 * pr.B));
        CK(hipMemset(pr.flag, 0, 4 * pr.B)); CK(hipMemset(pr.arrive, 0, 4 * pr.B)); CK(hipMemset(pr.census, 0, 4)); CK(hipMemset(pr.err, 0, 64)); CK(hipMemset(pr.cycles, 0, 8 * pr.B));
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
        unsigned err[16]; std::vector<u64> cyc(pr.B);
        CK(hipMemcpy(err, pr.err, 64, hipMemcpyDeviceToHost));
        if (err[4]) printf("first failing wait: site %x (kind %u, g %u, i %u) saw tags %u %u wanted %u\n", err[4], (err[4] >> 24) & 0x7f, (err[4] >> 16) & 0xff, err[4] & 0xffff, err[5], err[6], err[7]); CK(hipMemcpy(cyc.data(), pr.cycles, 8 * pr.B, hipMemcpyDeviceToHost));
        double mean = 0, mx = 0;
        for (int b = 0; b < pr.B; b++) { const double us = cyc[b] / 100.0 / (rounds - 10); mean += us / pr.B; mx = us > mx ? us : mx; }
        printf("B=%2d G=%d host=%d granules=%d xcdmap=%d(%u clusters on one XCD): %.2f us/round mean, %.2f max over clusters (wall %.1f ms)  timeouts %u mismatches %u/%u census_fail %u%s\n", pr.B, pr.G, pr.with_host, pr.granules, pr.xcd_map, err[8],
               mean, mx, wall_ms, err[0], err[1], err[2], err[3], host_tmo ? "  HOST TIMEOUT" : "");
        fflush(stdout);
        hipFree(pr.gpub); hipFree(pr.gpart); hipFree(pr.pub); hipFree(pr.part); hipFree(pr.flag); hipFree(pr.arrive); hipFree(pr.census); hipFree(pr.err); hipFree(pr.cycles); hipHostFree(hc); hipHostFree(hr); hipStreamDestroy(st);
    }
    return 0;
}
