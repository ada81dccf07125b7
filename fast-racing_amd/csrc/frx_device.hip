// hipcc translation unit: the gfx950 kernels and their launchers.
#include <hip/hip_runtime.h>

#include "frx_kernels.hpp"

namespace frx {

int launch_set_limits(const LaunchGeom &g) {
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void *)k_forward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_fwd)) != hipSuccess) return (int)e;
    if ((e = hipFuncSetAttribute((const void *)k_backward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bwd)) != hipSuccess) return (int)e;
    if ((e = hipFuncSetAttribute((const void *)k_penalty, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_pen)) != hipSuccess) return (int)e;
    if ((e = hipFuncSetAttribute((const void *)k_forward_knot, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_kfwd)) != hipSuccess) return (int)e;
    if ((e = hipFuncSetAttribute((const void *)k_backward_knot, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_kbwd)) != hipSuccess) return (int)e;
    return 0;
}
int launch_forward(const DevProblem &dp, const LaunchGeom &g, const double *x, double *T, double *C, double *band, void *stream) {
    if (g.solver == SOLVER_KNOT_PCR)
        hipLaunchKernelGGL(k_forward_knot, dim3(dp.B), dim3(g.knot_threads), g.lds_kfwd, (hipStream_t)stream, dp, x, T, C, g.maxCN, g.maxXb, g.maxVb);
    else
        hipLaunchKernelGGL(k_forward, dim3(dp.B), dim3(64), g.lds_fwd, (hipStream_t)stream, dp, x, T, C, band, g.maxN, g.maxCN);
    return (int)hipGetLastError();
}
int launch_penalty(const DevProblem &dp, const LaunchGeom &g, const double *T, const double *C, double *out20, void *stream) {
    hipLaunchKernelGGL(k_penalty, dim3((dp.P + g.ppw - 1) / g.ppw), dim3(64), g.lds_pen, (hipStream_t)stream, dp, T, C, out20, g.lpp,
                       g.ppw, g.Kmax);
    return (int)hipGetLastError();
}
int launch_backward(const DevProblem &dp, const LaunchGeom &g, const double *x, const double *T, const double *C,
                    const double *band, const double *out20, double *f, double *grad, void *stream) {
    if (g.solver == SOLVER_KNOT_PCR)
        hipLaunchKernelGGL(k_backward_knot, dim3(dp.B), dim3(g.knot_threads), g.lds_kbwd, (hipStream_t)stream, dp, x, T, C, out20, f,
                           grad, g.maxCN, g.maxXb, g.maxVb);
    else
        hipLaunchKernelGGL(k_backward, dim3(dp.B), dim3(64), g.lds_bwd, (hipStream_t)stream, dp, x, T, C, band, out20, f, grad, g.maxN,
                           g.maxCN);
    return (int)hipGetLastError();
}

} // namespace frx
