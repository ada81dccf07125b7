// hipcc translation unit: one workgroup per candidate, one launch per evaluation (frx_solo_kernel.hpp) and its launcher.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#define FRX_KERNEL_LINKAGE static          // the stage kernels of frx_kernels.hpp belong to frx_device.hip: here only their bodies are used
#include "frx_solo_kernel.hpp"

namespace frx {

static const void *solo_fn(int lpp) { return lpp == 17 ? (const void *)k_eval_solo<17> : lpp == 49 ? (const void *)k_eval_solo<49> : (const void *)k_eval_solo<0>; }

// fills g.lds_solo (0: this geometry keeps the stage kernels): candidates of <= 64 pieces on the knot solver, one quadrature sample per lane
int eval_solo_geometry(LaunchGeom &g, int samples_per_piece) {
    g.lds_solo = 0;
    // (one quadrature sample per lane: with kappa + 1 > 64 a lane of the stage kernel walks over several samples - k_penalty_lat's loop, not this kernel's passes)
    if (g.solver != SOLVER_KNOT_PCR || g.knot_threads != 64 || samples_per_piece > 64 || g.lpp != samples_per_piece || g.lpp < 1) return 0;
    if (g.maxXb + 16 > (int)SOLO_RB) return 0;                             // (the search direction of a round's tap is parked in the bodies' row buffer)
    const size_t lds = sizeof(double) * (size_t)solo_lds(g.maxN, g.maxXb, g.maxVb, g.maxCN, g.pcr_steps, 256 / g.lpp, g.Kmax).total;
    if (lds > (size_t)160 * 1024) return 0;
    g.lds_solo = lds;
    return 1;
}
// (see launch_set_limits, frx_device.hip: the dynamic-LDS limit of a kernel only grows, per device)
int eval_solo_raise_limit(const LaunchGeom &g) {
    static std::mutex mu;
    static size_t held[64][3] = {};
    if (!g.lds_solo) return 0;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    const int slot = g.lpp == 17 ? 0 : g.lpp == 49 ? 1 : 2;
    if (g.lds_solo <= held[dev][slot]) return 0;
    const hipError_t e = hipFuncSetAttribute(solo_fn(g.lpp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_solo);
    if (e == hipSuccess) held[dev][slot] = g.lds_solo;
    return (int)e;
}
int eval_solo_blocks_per_cu(const LaunchGeom &g) {
    int n = 0;
    if (!g.lds_solo) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, solo_fn(g.lpp), 256, g.lds_solo) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int launch_eval_solo(const DevProblem &dp, const LaunchGeom &g, const double *x, double *T, double *C, double *out20, double *f, double *grad, void *stream,
                     const double *tap_d, const int *tap_flags, void *tap_res, unsigned *tap_arrive, volatile unsigned *tap_flag, unsigned tap_round) {
    if (!g.lds_solo) return (int)hipErrorInvalidValue;
    SoloArgs a;
    a.x = x; a.T = T; a.C = C; a.out20 = out20; a.f = f; a.g = grad; a.pcrw = g.pcrw;
    a.maxCN = g.maxCN; a.maxXb = g.maxXb; a.maxVb = g.maxVb; a.nsteps = g.pcr_steps; a.lpp = g.lpp; a.ppg = 256 / g.lpp; a.Kmax = g.Kmax; a.maxN = g.maxN;
    const LineSearchTap tap{tap_d, tap_flags, (DvResult *)tap_res, tap_arrive, tap_flag, tap_round};
    if (g.lpp == 17) hipLaunchKernelGGL(k_eval_solo<17>, dim3(dp.B), dim3(256), g.lds_solo, (hipStream_t)stream, dp, a, tap);
    else if (g.lpp == 49) hipLaunchKernelGGL(k_eval_solo<49>, dim3(dp.B), dim3(256), g.lds_solo, (hipStream_t)stream, dp, a, tap);
    else hipLaunchKernelGGL(k_eval_solo<0>, dim3(dp.B), dim3(256), g.lds_solo, (hipStream_t)stream, dp, a, tap);
    return (int)hipGetLastError();
}
} // namespace frx
