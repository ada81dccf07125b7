// hipcc translation unit: the one-launch evaluation (frx_eval_kernel.hpp) and its launcher.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

#define FRX_KERNEL_LINKAGE static          // the stage kernels of frx_kernels.hpp belong to frx_device.hip: here only their bodies are used
#include "frx_eval_kernel.hpp"

namespace frx {

static bool eval_argp() { static const bool on = [] { const char *e = std::getenv("FRX_EVAL_ARGPTR"); return !(e && e[0] == '0'); }(); return on; }   // FRX_EVAL_ARGPTR=0: the by-value form (A/B)

// (see launch_set_limits, frx_device.hip: the dynamic-LDS limit of a kernel only grows, per device)
int eval_cluster_raise_limit(size_t bytes) {
    static std::mutex mu;
    static size_t held[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (bytes <= held[dev]) return 0;
    const hipError_t e = hipFuncSetAttribute(eval_argp() ? (const void *)k_eval_cluster<true> : (const void *)k_eval_cluster<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
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
    const void *fn = eval_argp() ? (const void *)k_eval_cluster<true> : (const void *)k_eval_cluster<false>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
size_t eval_cluster_args_bytes() { return sizeof(EvalClusterArgs); }
// The handle's constant arguments (frx_api.cpp keeps a host copy next to a device copy and uploads it when it changes - at create, and when a diagnostic switches the
// cycle stamps on): everything launch_eval_cluster used to pack per call.
void eval_cluster_args(const DevProblem &dp, const LaunchGeom &g, double *T, double *C, unsigned long long *ll, unsigned *words, void *out) {
    EvalClusterArgs a;
    std::memset(&a, 0, sizeof(a));
    a.dp = dp; a.T = T; a.C = C; a.out20ll = ll; a.ctll = ll + (size_t)40 * dp.P; a.words = words; a.status = words + (size_t)64 * dp.B;
    a.G = g.ev_G; a.maxCN = g.maxCN; a.maxXb = g.maxXb; a.maxVb = g.maxVb; a.nsteps = g.pcr_steps; a.lpp = g.lpp; a.ppw = g.ppw; a.Kmax = g.Kmax; a.pen_lds = eval_pen_lds(g); a.maxN19 = g.maxN * 19;
    std::memcpy(out, &a, sizeof(a));
}
int launch_eval_cluster(const LaunchGeom &g, int B, const void *args_host, const void *args_dev, const double *x, double *f, double *grad,
                        unsigned long long timeout_ticks, void *stream, unsigned *status_host) {
    if (!g.ev_G) return (int)hipErrorInvalidValue;
    EvalCallArgs c;
    c.x = x; c.f = f; c.g = grad; c.status_host = status_host; c.timeout_ticks = timeout_ticks;
    { const char *e = std::getenv("FRX_EVAL_FUSED_WT"); c.force_wt = (e && e[0] == '1') ? 1 : 0; }
    c.test_drop_members = timeout_ticks == 1ull ? 1 : 0;               // (test mode, frx_debug_set_eval_fused(p, 2): members that never arrive and a 50 us bound)
    if (c.test_drop_members) c.timeout_ticks = 5000ull;
    const dim3 grid(8 * g.ev_G * ((B + 7) / 8));
    if (eval_argp() && args_dev) hipLaunchKernelGGL(k_eval_cluster<true>, grid, dim3(256), g.lds_ev, (hipStream_t)stream, (const EvalClusterArgs *)args_dev, c);
    else hipLaunchKernelGGL(k_eval_cluster<false>, grid, dim3(256), g.lds_ev, (hipStream_t)stream, *(const EvalClusterArgs *)args_host, c);
    return (int)hipGetLastError();
}
} // namespace frx
