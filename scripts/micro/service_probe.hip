// What would a RESIDENT evaluation service cost per call ("measure their latency first ... if a memory op costs more than the 3.1 us it saves, record
// that and stop")?  The service keeps the clusters' leaders on the chip; a stream-ordered call has to (a) ring them and (b) return when they are done.  Two ways to put
// that on a stream, both timed here with NO work in the service (the protocol's floor), 32 resident leaders as at the headline batch:
//   kernel:  a one-workgroup doorbell kernel per call - lane k stores the call's tag into leader k's doorbell (write-through), polls leader k's `done` word (L1-bypassing
//            loads), leaves; K calls as K dependent launches, direct and as one hipGraph
//   memops:  hipStreamWriteValue64 (doorbell) + hipStreamWaitValue64 (done), per call, one doorbell for all leaders
// Build: hipcc --offload-arch=gfx950 -O3 -o service_probe service_probe.hip     Run: ./service_probe [K]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;

__global__ void k_service(u64 *bell, u64 *done, int shared_bell, u64 quit) {          // leader k = block k: waits for tag t = 1, 2, ... on its doorbell, answers at once
    const int k = blockIdx.x;
    u64 *b = bell + (shared_bell ? 0 : 16 * k), *d = done + 16 * k;
    if (threadIdx.x != 0) return;
    const u64 t_end = wall_clock64() + 1000000000ull;                                  // every spin of this probe is bounded: 10 s of the 100 MHz counter
    for (u64 t = 1;; t++) {
        u64 v;
        for (unsigned spins = 0;; spins++) {
            v = __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v >= t) break;
            if ((spins & 1023u) == 1023u && wall_clock64() > t_end) return;
            __builtin_amdgcn_s_sleep(1);
        }
        if (v == quit) return;
        __hip_atomic_store(d, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_ring(u64 *bell, u64 *done, u64 *tagp, int n) {                         // one call: ring n leaders, wait for all of them
    const int k = threadIdx.x;
    const u64 t = *tagp + 1;
    if (k < n) {
        __hip_atomic_store(bell + 16 * k, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 t_end = wall_clock64() + 200000000ull;
        for (unsigned spins = 0; __hip_atomic_load(done + 16 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t; spins++) {
            if ((spins & 1023u) == 1023u && wall_clock64() > t_end) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (k == 0) *tagp = t;                                                               // (device-side call counter: a captured graph replays the same kernel arguments)
}
__global__ void k_empty(u64 *p) { if (p && threadIdx.x == 1000) *p = 1; }

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 200, L = 32;
    u64 *bell, *done, *tagp;
    CK(hipMalloc(&bell, 16 * 8 * L)); CK(hipMalloc(&done, 16 * 8 * L)); CK(hipMalloc(&tagp, 8));
    hipStream_t svc, s;
    CK(hipStreamCreateWithFlags(&svc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset = [&]() { (void)hipMemset(bell, 0, 16 * 8 * L); (void)hipMemset(done, 0, 16 * 8 * L); (void)hipMemset(tagp, 0, 8); return hipDeviceSynchronize(); };
    float ms = 0.f;
    // 0. the floor of K dependent EMPTY launches (direct and as a graph): what a stream-ordered call costs before it does anything
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (u64 *)nullptr);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (u64 *)nullptr); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("{\"what\": \"K dependent empty one-workgroup launches, direct\", \"K\": %d, \"us_per_call\": %.3f}\n", K, 1e3 * ms / K);
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (u64 *)nullptr);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 2; w++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("{\"what\": \"the same as ONE hipGraph\", \"K\": %d, \"us_per_call\": %.3f}\n", K, 1e3 * ms / K);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    // 1. doorbell kernel per call against 32 resident leaders
    CK(reset());
    hipLaunchKernelGGL(k_service, dim3(L), dim3(64), 0, svc, bell, done, 0, ~0ull);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_ring, dim3(1), dim3(64), 0, s, bell, done, tagp, L);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_ring, dim3(1), dim3(64), 0, s, bell, done, tagp, L); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("{\"what\": \"doorbell kernel per call, 32 resident leaders, empty service, direct launches\", \"K\": %d, \"us_per_call\": %.3f}\n", K, 1e3 * ms / K);
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_ring, dim3(1), dim3(64), 0, s, bell, done, tagp, L);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 2; w++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("{\"what\": \"doorbell kernel per call, as ONE hipGraph\", \"K\": %d, \"us_per_call\": %.3f}\n", K, 1e3 * ms / K);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    {   // quit the leaders: every doorbell gets the quit value
        std::vector<u64> q(16 * L, ~0ull);
        CK(hipMemcpyAsync(bell, q.data(), 16 * 8 * L, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(svc));
    }
    // 2. stream memory operations: write the doorbell, wait for the LAST leader's done word (all leaders share the doorbell; leader L-1 answers like the others)
    {
        u64 *sig = nullptr;
        hipError_t e = hipExtMallocWithFlags((void **)&sig, 16 * 8 * (L + 1), hipMallocSignalMemory);
        if (e != hipSuccess) { (void)hipGetLastError(); printf("{\"what\": \"stream memory operations\", \"error\": \"hipExtMallocWithFlags(hipMallocSignalMemory): %s\"}\n", hipGetErrorString(e)); }
        else {
            // signal memory is 8 bytes per allocation on some runtimes: use ONE word pair through plain device memory for the doorbell and the signal word for `done`
            (void)hipMemset(sig, 0, 8); CK(reset());
            hipLaunchKernelGGL(k_service, dim3(1), dim3(64), 0, svc, bell, sig, 1, ~0ull);     // one leader, doorbell = bell[0], done = sig[0]
            hipError_t ew = hipSuccess, eq = hipSuccess;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= 20 && ew == hipSuccess && eq == hipSuccess; i++) { ew = hipStreamWriteValue64(s, bell, (u64)i, 0); eq = hipStreamWaitValue64(s, sig, (u64)i, hipStreamWaitValueGte, ~0ull); }
            if (ew != hipSuccess || eq != hipSuccess) printf("{\"what\": \"stream memory operations\", \"error\": \"write: %s, wait: %s\"}\n", hipGetErrorString(ew), hipGetErrorString(eq));
            else {
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int i = 21; i <= 20 + K; i++) { (void)hipStreamWriteValue64(s, bell, (u64)i, 0); (void)hipStreamWaitValue64(s, sig, (u64)i, hipStreamWaitValueGte, ~0ull); }
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("{\"what\": \"hipStreamWriteValue64 + hipStreamWaitValue64 per call, one resident leader, empty service\", \"K\": %d, \"us_per_call\": %.3f, \"host_wall_us_per_call\": %.3f}\n", K, 1e3 * ms / K,
                       std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (K + 20));
            }
            u64 q = ~0ull; (void)hipMemcpyAsync(bell, &q, 8, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(svc);
        }
    }
    printf("{\"what\": \"reference points\", \"cold_start_of_k_eval_cluster_us\": 3.1, \"one_launch_evaluation_us\": 16.7, \"the_same_bodies_resident_in_k_round_us\": 13.5}\n");
    return 0;
}
