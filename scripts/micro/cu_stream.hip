// Micro-benchmark: how fast can ONE workgroup (one CU) stream from HBM?  B workgroups, each reading its own contiguous chunk with
// W waves, U independent 16-byte loads per lane per trip.  Prints GB/s per workgroup.  (Sizing aid for k_lbfgs_pre.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int U>
__global__ void k_stream(const double2 *__restrict__ src, double *out, size_t per_wg, int trips) {
    const double2 *p = src + (size_t)blockIdx.x * per_wg + threadIdx.x;
    const int stride = blockDim.x;
    double acc = 0.0;
    for (int t = 0; t < trips; t++) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = p[(size_t)(t * U + u) * stride];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x + v[u].y;
    }
    if (acc == 123.456) out[blockIdx.x] = acc;
}
int main() {
    const int B = 32;
    const size_t per_wg_bytes = 8u << 20;            // 8 MB per workgroup, 256 MB total: no reuse
    double2 *src; double *out;
    hipMalloc(&src, per_wg_bytes * B); hipMalloc(&out, 8 * B);
    hipMemset(src, 0, per_wg_bytes * B);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int W : {1, 2, 3, 4, 6, 8, 12, 16}) {
        for (int U : {4, 8, 16, 32}) {
            const int threads = 64 * W;
            const size_t per_wg = per_wg_bytes / 16;
            const int trips = (int)(per_wg / ((size_t)threads * U));
            auto launch = [&]() {
                if (U == 4) hipLaunchKernelGGL(k_stream<4>, dim3(B), dim3(threads), 0, 0, src, out, per_wg, trips);
                if (U == 8) hipLaunchKernelGGL(k_stream<8>, dim3(B), dim3(threads), 0, 0, src, out, per_wg, trips);
                if (U == 16) hipLaunchKernelGGL(k_stream<16>, dim3(B), dim3(threads), 0, 0, src, out, per_wg, trips);
                if (U == 32) hipLaunchKernelGGL(k_stream<32>, dim3(B), dim3(threads), 0, 0, src, out, per_wg, trips);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("W=%2d U=%2d in-flight/CU %4d KB: %.1f us, %.1f GB/s per workgroup\n", W, U, W * U, ms * 1e3, per_wg_bytes / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
