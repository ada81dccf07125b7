// hipcc translation unit: the one-launch evaluation (frx_eval_kernel.hpp) and its launcher.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#define FRX_KERNEL_LINKAGE static          // the stage kernels of frx_kernels.hpp belong to frx_device.hip: here only their bodies are used
#include "frx_eval_kernel.hpp"

namespace frx {

// (see launch_set_limits, frx_device.hip: the dynamic-LDS limit of a kernel only grows, per device)
int eval_cluster_raise_limit(size_t bytes) {
    static std::mutex mu;
    static size_t held[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (bytes <= held[dev]) return 0;
    const hipError_t e = hipFuncSetAttribute((const void *)k_eval_cluster, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) held[dev] = bytes;
    return (int)e;
}
static int eval_pen_lds(const LaunchGeom &g) { return g.ppw * 19 + g.ppw * (g.Kmax + 1) * 4 + 64 * 21; }   // doubles per wave (penalty_body with a 64-lane group)
int eval_cluster_geometry(LaunchGeom &g) {
    g.ev_G = 0; g.lds_ev = 0;
    if (g.solver != SOLVER_KNOT_PCR || g.knot_threads != 64 || g.ppw < 1) return 0;
    const int ntasks = (g.maxN + g.ppw - 1) / g.ppw;
    const size_t lds = sizeof(double) * (size_t)eval_cluster_lds(g.maxN * 19, g.maxXb, g.maxVb, g.maxCN, g.pcr_steps, eval_pen_lds(g)).total;
    if (lds > (size_t)160 * 1024) return 0;
    g.ev_G = 1 + (ntasks + 3) / 4;                                   // the members take every wave-task of the largest candidate in one pass
    g.lds_ev = lds;
    return g.ev_G;
}
int eval_cluster_blocks_per_cu(size_t lds_bytes) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_eval_cluster, 256, lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int launch_eval_cluster(const DevProblem &dp, const LaunchGeom &g, const double *x, double *T, double *C, double *f, double *grad,
                        unsigned long long *ll, unsigned *words, unsigned long long timeout_ticks, void *stream, unsigned *status_host) {
    if (!g.ev_G) return (int)hipErrorInvalidValue;
    EvalClusterArgs a;
    a.status_host = status_host;
    a.dp = dp; a.x = x; a.T = T; a.C = C; a.f = f; a.g = grad; a.out20ll = ll; a.ctll = ll + (size_t)40 * dp.P; a.words = words; a.status = words + (size_t)64 * dp.B; a.timeout_ticks = timeout_ticks;
    a.G = g.ev_G; a.maxCN = g.maxCN; a.maxXb = g.maxXb; a.maxVb = g.maxVb; a.nsteps = g.pcr_steps; a.lpp = g.lpp; a.ppw = g.ppw; a.Kmax = g.Kmax; a.pen_lds = eval_pen_lds(g); a.maxN19 = g.maxN * 19;
    { const char *e = std::getenv("FRX_EVAL_FUSED_WT"); a.force_wt = (e && e[0] == '1') ? 1 : 0; }
    a.test_drop_members = timeout_ticks == 1ull ? 1 : 0;               // (test mode, frx_debug_set_eval_fused(p, 2): members that never arrive and a 50 us bound)
    if (a.test_drop_members) a.timeout_ticks = 5000ull;
    hipLaunchKernelGGL(k_eval_cluster, dim3(8 * g.ev_G * ((dp.B + 7) / 8)), dim3(256), g.lds_ev, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
} // namespace frx
