// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the knot kernels use: a coalesced copy of a known
// byte count with 8-byte and with 16-byte accesses per lane (MI355X_MICROARCH.md, "HBM": FETCH_SIZE reports half the bytes of a 16-byte-per-lane
// stream; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; the kernels copy BYTES bytes each, 5 launches.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_copy8(const double *__restrict__ a, double *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_copy16(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const size_t BYTES = (size_t)64 << 20;
    double *a, *b;
    if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&b, BYTES) != hipSuccess) return 1;
    hipMemset(a, 1, BYTES); hipMemset(b, 0, BYTES);
    for (int r = 0; r < 5; r++) {
        hipLaunchKernelGGL(k_copy8, dim3(4096), dim3(256), 0, 0, a, b, BYTES / 8);
        hipLaunchKernelGGL(k_copy16, dim3(4096), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, BYTES / 16);
    }
    hipDeviceSynchronize();
    printf("copied %zu bytes per launch\n", BYTES);
    return 0;
}
