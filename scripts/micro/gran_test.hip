// minimal check of the tag-in-payload hand-off between two workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__global__ void k(u64 *buf, u64 *out, int rounds, int n, int mode) {
    const int t = threadIdx.x, g = blockIdx.x;
    for (int r = 1; r <= rounds; r++) {
        if (g == 0) {
            // wait for ack of previous round
            if (t == 0 && r > 1) { u64 dl = wall_clock64() + 100000000ull; while (__hip_atomic_load(buf + 4096, RLX_AGENT) != (u64)(r - 1) && wall_clock64() < dl) {} }
            __syncthreads();
            for (int i = t; i < n; i += blockDim.x) __hip_atomic_store(buf + i, ((u64)r << 32) | (unsigned)(i * 7 + r), RLX_AGENT);
        } else {
            int bad = 0; u64 last = 0;
            for (int i = t; i < n; i += blockDim.x) {
                u64 dl = wall_clock64() + 100000000ull, a = 0;
                for (;;) {
                    if (mode == 0) a = __hip_atomic_load(buf + i, RLX_AGENT);
                    else a = __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((a >> 32) == (u64)r) break;
                    if (wall_clock64() > dl) { bad++; break; }
                }
                last = a;
            }
            if (bad) { atomicAdd((unsigned long long *)out, (u64)bad); out[1 + (t & 7)] = last; }
            __syncthreads();
            if (t == 0) __hip_atomic_store(buf + 4096, (u64)r, RLX_AGENT);
        }
    }
}
int main() {
    u64 *buf, *out; hipMalloc(&buf, 8 * 8192); hipMalloc(&out, 8 * 16);
    for (int mode = 0; mode < 2; mode++) for (int n : {1, 64, 704}) {
        hipMemset(buf, 0, 8 * 8192); hipMemset(out, 0, 8 * 16);
        hipLaunchKernelGGL(k, dim3(2), dim3(256), 0, 0, buf, out, 200, n, mode);
        hipDeviceSynchronize();
        u64 h[16]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d n %d: failures %llu last seen %llx %llx\n", mode, n, h[0], h[1], h[2]);
    }
    return 0;
}
