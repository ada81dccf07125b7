// hipcc translation unit: the gfx950 kernels and their launchers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "frx_kernels.hpp"
#include "frx_lbfgs_kernels.hpp"
#include "frx_round_kernel.hpp"
#include "frx_eval_kernel.hpp"
#include "frx_corridor_kernels.hpp"

namespace frx {

// The dynamic-LDS limit of a kernel is a property of the FUNCTION, not of a handle: handles of different geometry live side by side (tests, a planner with several
// corridors), so the limit only ever grows - a handle created later with a smaller need must not lower it under an older handle's launches.
static int raise_lds_limit(const void *fn, size_t bytes, size_t &held) {
    if (bytes <= held) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) held = bytes;
    return (int)e;
}
int launch_set_limits(const LaunchGeom &g) {
    static std::mutex mu;
    static size_t held_all[64][10] = {};                             // per device: a function's attributes belong to the device that is current
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    size_t *held = held_all[dev];
    int e;
    if ((e = raise_lds_limit((const void *)k_forward, g.lds_fwd, held[0]))) return e;
    if ((e = raise_lds_limit((const void *)k_backward, g.lds_bwd, held[1]))) return e;
    if ((e = raise_lds_limit((const void *)k_penalty, g.lds_pen, held[2]))) return e;
    if ((e = raise_lds_limit((const void *)k_penalty_lat, g.lds_pen, held[3]))) return e;
    if (g.lds_pen2 && (e = raise_lds_limit((const void *)k_penalty_lat2, g.lds_pen2, held[4]))) return e;
    if ((e = raise_lds_limit((const void *)k_forward_knot, g.lds_kfwd, held[5]))) return e;
    if ((e = raise_lds_limit((const void *)k_backward_knot, g.lds_kbwd, held[6]))) return e;
    if ((e = raise_lds_limit((const void *)k_forward_knot64, g.lds_kfwd, held[7]))) return e;
    if ((e = raise_lds_limit((const void *)k_backward_knot64, g.lds_kbwd, held[8]))) return e;
    if (g.ev_G && (e = raise_lds_limit((const void *)k_eval_cluster, g.lds_ev, held[9]))) return e;
    return 0;
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
int launch_forward(const DevProblem &dp, const LaunchGeom &g, const double *x, double *T, double *C, double *band, void *stream) {
    if (g.solver == SOLVER_KNOT_PCR)
        if (g.knot_threads == 64) hipLaunchKernelGGL(k_forward_knot64, dim3(dp.B), dim3(256), g.lds_kfwd, (hipStream_t)stream, dp, x, T, C, g.maxCN, g.maxXb, g.maxVb, g.pcrw, g.pcr_steps);
        else hipLaunchKernelGGL(k_forward_knot, dim3(dp.B), dim3(256), g.lds_kfwd, (hipStream_t)stream, dp, x, T, C, g.maxCN, g.maxXb, g.maxVb, g.knot_threads, g.pcrw, g.pcr_steps);
    else
        hipLaunchKernelGGL(k_forward, dim3(dp.B), dim3(64), g.lds_fwd, (hipStream_t)stream, dp, x, T, C, band, g.maxN, g.maxCN);
    return (int)hipGetLastError();
}
int launch_penalty(const DevProblem &dp, const LaunchGeom &g, const double *T, const double *C, double *out20, void *stream) {
    // default: the latency form (148 VGPRs, 3 waves per SIMD, phases interleaved by the scheduler) - measured faster at the headline batch
    // (4.97 vs 5.23 us) AND at 1024 candidates (39.8 vs 42.3 us) than the throughput form (126 VGPRs, 4 waves per SIMD); FRX_PENALTY_FORM=thr selects that one
    static const int forced = [] { const char *e = std::getenv("FRX_PENALTY_FORM"); return !e ? 0 : e[0] == 'l' ? 1 : 2; }();
    const bool lat = forced != 2;
    const int nwg = (dp.P + g.ppg - 1) / g.ppg;
    const char *tp_env = std::getenv("FRX_PENALTY_TWOPHASE");           // (read per launch: the test toggles it inside one process)
    const bool two_phase = !(tp_env && tp_env[0] == '0');
    if (lat && two_phase && g.lds_pen2) hipLaunchKernelGGL(k_penalty_lat2, dim3(nwg), dim3(64 * g.pen_w), g.lds_pen2, (hipStream_t)stream, dp, T, C, out20, g.lpp, g.ppg, g.Kmax);
    else if (lat) hipLaunchKernelGGL(k_penalty_lat, dim3(nwg), dim3(64 * g.pen_w), g.lds_pen, (hipStream_t)stream, dp, T, C, out20, g.lpp, g.ppg, g.Kmax);
    else hipLaunchKernelGGL(k_penalty, dim3(nwg), dim3(64 * g.pen_w), g.lds_pen, (hipStream_t)stream, dp, T, C, out20, g.lpp, g.ppg, g.Kmax);
    return (int)hipGetLastError();
}
int launch_backward(const DevProblem &dp, const LaunchGeom &g, const double *x, const double *T, const double *C,
                    const double *band, const double *out20, double *f, double *grad, void *stream, const double *tap_d, const int *tap_flags,
                    void *tap_res, unsigned *tap_arrive, volatile unsigned *tap_flag, unsigned tap_round) {
    if (g.solver == SOLVER_KNOT_PCR && g.knot_threads == 64)
        hipLaunchKernelGGL(k_backward_knot64, dim3(dp.B), dim3(256), g.lds_kbwd, (hipStream_t)stream, dp, x, T, C, out20, f,
                           grad, g.maxCN, g.maxXb, g.maxVb, g.pcrw, g.pcr_steps,
                           LineSearchTap{tap_d, tap_flags, (DvResult *)tap_res, tap_arrive, tap_flag, tap_round});
    else if (g.solver == SOLVER_KNOT_PCR)
        hipLaunchKernelGGL(k_backward_knot, dim3(dp.B), dim3(256), g.lds_kbwd, (hipStream_t)stream, dp, x, T, C, out20, f,
                           grad, g.maxCN, g.maxXb, g.maxVb, g.knot_threads, g.pcrw, g.pcr_steps,
                           LineSearchTap{tap_d, tap_flags, (DvResult *)tap_res, tap_arrive, tap_flag, tap_round});
    else
        hipLaunchKernelGGL(k_backward, dim3(dp.B), dim3(64), g.lds_bwd, (hipStream_t)stream, dp, x, T, C, band, out20, f, grad, g.maxN,
                           g.maxCN);
    return (int)hipGetLastError();
}


static DvBuffers to_buffers(const DvLaunch &dv) {
    DvBuffers b;
    b.xoff = dv.xoff; b.x = dv.x; b.g = dv.g; b.xp = dv.xp; b.gp = dv.gp; b.d = dv.d; b.S = dv.S; b.Y = dv.Y; b.ys = dv.ys; b.gt = dv.gt; b.dflags = dv.dflags; b.pflags = dv.pflags; b.poff = dv.poff; b.m = dv.m; b.B = dv.B;
    return b;
}
int launch_lbfgs_pre(const DvLaunch &dv, const void *cmd, void *res, void *stream) {
    const DvBuffers b = to_buffers(dv);
    const DvCommand *c = (const DvCommand *)cmd; DvResult *r = (DvResult *)res;
    hipStream_t st = (hipStream_t)stream;
#define FRX_PRE(E_, W_, PF_, BLK_) case ((E_ * 16 + W_) * 64 + PF_) * 8 + BLK_: hipLaunchKernelGGL((k_lbfgs_pre<E_, W_, PF_, BLK_>), dim3(dv.B), dim3(64 * W_), 0, st, b, c, r); break;
    switch (((dv.E * 16 + dv.W) * 64 + dv.PF) * 8 + dv.BLK) {
    FRX_PRE(2, 1, 16, 4) FRX_PRE(2, 2, 16, 4) FRX_PRE(2, 3, 16, 4) FRX_PRE(2, 4, 16, 4) FRX_PRE(2, 5, 16, 4) FRX_PRE(2, 6, 16, 4) FRX_PRE(2, 7, 16, 4) FRX_PRE(2, 8, 16, 4)
    FRX_PRE(4, 1, 8, 4) FRX_PRE(4, 2, 8, 4) FRX_PRE(4, 3, 8, 4) FRX_PRE(4, 4, 8, 4) FRX_PRE(4, 5, 8, 4) FRX_PRE(4, 6, 8, 4) FRX_PRE(4, 7, 8, 4) FRX_PRE(4, 8, 8, 4)
    FRX_PRE(6, 1, 8, 4) FRX_PRE(6, 2, 8, 4) FRX_PRE(6, 3, 8, 4) FRX_PRE(6, 4, 8, 4) FRX_PRE(6, 5, 8, 4) FRX_PRE(6, 6, 8, 4) FRX_PRE(6, 7, 8, 4) FRX_PRE(6, 8, 8, 4)
    FRX_PRE(8, 1, 4, 4) FRX_PRE(8, 2, 4, 4) FRX_PRE(8, 3, 4, 4) FRX_PRE(8, 4, 4, 4) FRX_PRE(8, 5, 4, 4) FRX_PRE(8, 6, 4, 4) FRX_PRE(8, 7, 4, 4) FRX_PRE(8, 8, 4, 4)
    FRX_PRE(4, 3, 8, 1) FRX_PRE(4, 3, 8, 2) FRX_PRE(2, 6, 16, 1)     // experiments (FRX_DV_GEOM)
    default: return (int)hipErrorInvalidValue;
    }
#undef FRX_PRE
    return (int)hipGetLastError();
}
int launch_lbfgs_post(const DvLaunch &dv, const double *f, const void *cmd, void *res, void *stream) {
    hipLaunchKernelGGL(k_lbfgs_post, dim3(dv.B), dim3(64), 0, (hipStream_t)stream, to_buffers(dv), f, (const DvCommand *)cmd, (DvResult *)res);
    return (int)hipGetLastError();
}

// leader's resident LDS operands: (C, T) copy, x, waypoint polytopes, direction, reduction multipliers (ResidentOps), then gradient,
// previous point and previous gradient (rk_leader_loop)
static int round_ct_doubles(const LaunchGeom &g) {
    const int xpad = (g.maxXb + 1) & ~1, vpad = (g.maxVb + g.knot_threads + 1) & ~1, pw = (g.pcr_steps * 8 + 5) * g.knot_threads;
    return ((g.maxN * 19 + 1) & ~1) + 5 * xpad + vpad + ((pw + 1) & ~1) + 4 * g.knot_threads;
}
static int round_eval_doubles(const LaunchGeom &g) {
    const size_t pen = (size_t)g.ppw * 19 + (size_t)g.ppw * (g.Kmax + 1) * 4 + 64 * 21;              // doubles per wave (LaunchGeom::lds_pen)
    size_t e = std::max(g.lds_kfwd, g.lds_kbwd) / sizeof(double) + 2;
    // <= 64 pieces: the evaluation bodies find x, polytopes, direction and multipliers in the resident operands, so their scratch ends
    // behind the knot arrays (rows | knot arrays | Tf, gT | gCo | cross-wave partials)
    if (g.knot_threads == 64) e = (size_t)36 * 64 + 9 * 65 + 2 * 64 + g.maxCN + 16;
    e = std::max(e, 4 * pen + 8);
    return (int)((e + 1) & ~(size_t)1) + round_ct_doubles(g);
}
size_t round_lds_bytes(const LaunchGeom &g, int m, int E) {
    if ((E != ROUND_E && E != ROUND_E_SMALL) || m < 1 || m > 128 || g.solver != SOLVER_KNOT_PCR) return 0;
    return sizeof(double) * (size_t)round_lds(m, 2 * E, round_eval_doubles(g)).total;
}
static bool n64_class(const LaunchGeom &g) { static const bool generic = [] { const char *e = std::getenv("FRX_RESIDENT_NR"); return e && e[0] == '0'; }(); return g.knot_threads == 64 && !generic; }   // FRX_RESIDENT_NR=0: the generic instantiation (A/B)
int launch_round(const DevProblem &dp, const LaunchGeom &g, const RoundLaunch &r, void *stream) {
    if (r.E != ROUND_E && r.E != ROUND_E_SMALL) return (int)hipErrorInvalidValue;
    RoundArgs a;
    a.dp = dp;
    a.maxCN = g.maxCN; a.maxXb = g.maxXb; a.maxVb = g.maxVb; a.nrow = g.knot_threads; a.nsteps = g.pcr_steps; a.lpp = g.lpp; a.ppw = g.ppw; a.Kmax = g.Kmax;
    a.pen_lds = g.ppw * 19 + g.ppw * (g.Kmax + 1) * 4 + 64 * 21;
    a.x = r.x; a.g = r.g; a.xp = r.xp; a.gp = r.gp; a.d = r.d; a.f = r.f; a.T = r.T; a.C = r.C; a.out20 = r.out20; a.pcrw = g.pcrw;
    a.out20ll = g.knot_threads == 64 ? r.out20ll : nullptr;                                                      // (only the <= 64-piece adjoint polls granules)
    a.pubsyg = r.pubsyg; a.part = r.part; a.upub = r.upub; a.dpub = r.dpub; a.dbg = r.dbg; a.dbg_cap = r.dbg_cap; a.dbg_cands = r.dbg_cands;
    a.phase = r.words; a.cntA = r.words + 32; a.uflag = r.words + 64; a.cntL = r.words + 96;                     // one 512-byte block per candidate, one 128-byte line per word
    if (r.S < 1 || r.S > r.B) return (int)hipErrorInvalidValue;
    a.census = r.words + (size_t)RK_WORDS_PER_CAND * r.S; a.status = a.census + 1; a.xcc = a.census + 2; a.spec = a.xcc + (size_t)r.S * r.G;
    a.h_cmd = (RoundCmd *)r.h_cmd; a.h_res = (RoundRes *)r.h_res;
    a.timeout_ticks = r.timeout_ticks;
    a.census_ticks = std::min<unsigned long long>(r.timeout_ticks, 25000000ull);           // 250 ms
    a.ls_ftol = r.ls_ftol; a.ls_gtol = r.ls_gtol; a.ls_min_step = r.ls_min_step; a.ls_max_step = r.ls_max_step; a.ls_xtol = r.ls_xtol; a.ls_max_linesearch = r.ls_max_linesearch; a.speculate = r.speculate;
    { static const int ps = [] { const char *e = std::getenv("FRX_RESIDENT_POLL"); return e ? std::atoi(e) : 0; }(); a.poll_sleep = ps < 0 ? 0 : ps > 4 ? 4 : ps; }
    a.cmd_stride = r.cmd_stride; a.stamp_round = r.stamp_round; a.fast_control = r.fast_control;
    a.B = r.B; a.S = r.S; a.G = r.G; a.m = r.m; a.NXP = r.NXP; a.eval_doubles = round_eval_doubles(g); a.ct_doubles = round_ct_doubles(g); a.maxN19 = g.maxN * 19;
    const size_t lds = round_lds_bytes(g, r.m, r.E);
    a.prof = (rk_u64 *)r.prof;
    a.trace = r.prof ? (rk_u64 *)r.trace : nullptr; a.trace_cap = r.trace_cap; a.trace_lo = r.trace_lo; a.trace_hi = r.trace_hi;
    // Instantiations: history elements per thread (56: six history workgroups at the headline size; 28: twelve, for batches that leave the chip room -
    // no history register in an AGPR, both history loops half as long), with / without the profile, for <= 64 pieces per candidate (the
    // class-specific bodies only) or any geometry.
    const bool n64 = n64_class(g);
    const dim3 grid(8 * r.G * ((r.S + 7) / 8)), block(256);
    auto go = [&](auto kernel) -> int {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)stream, a);
        return (int)hipGetLastError();
    };
    a.rs.cand = r.rs_cand; a.rs.f_last = r.rs_f; a.rs.S = r.rs_S; a.rs.Y = r.rs_Y; a.rs.hs = r.rs_hs; a.rs.newest = r.rs_newest; a.rs.bound = r.rs_bound; a.rs.rinv = r.rs_rinv; a.rs.yy = r.rs_yy; a.rs.vd = r.rs_vd;
    if (r.rs_cand) {                                                   // take-over instantiation: <= 64 pieces, no profile
        if (!n64 || r.prof) return (int)hipErrorInvalidValue;
        return r.E == ROUND_E ? go(k_round<ROUND_E, false, 64, true>) : go(k_round<ROUND_E_SMALL, false, 64, true>);
    }
    if (r.E == ROUND_E) {
        if (r.prof) return n64 ? go(k_round<ROUND_E, true, 64>) : go(k_round<ROUND_E, true, 0>);
        return n64 ? go(k_round<ROUND_E, false, 64>) : go(k_round<ROUND_E, false, 0>);
    }
    if (r.prof) return n64 ? go(k_round<ROUND_E_SMALL, true, 64>) : go(k_round<ROUND_E_SMALL, true, 0>);
    return n64 ? go(k_round<ROUND_E_SMALL, false, 64>) : go(k_round<ROUND_E_SMALL, false, 0>);
}

size_t dilate_lds_bytes(int pcap) { return sizeof(double) * ((size_t)3 * pcap + 32 + 36 + 16) + sizeof(int) * ((size_t)2 * pcap + 257 + 3); }
int launch_dilate(const DilateLaunch &d, void *stream) {
    DilateArgs a;
    a.p1 = d.p1; a.p2 = d.p2; a.obs = d.obs; a.bbox[0] = d.bbox[0]; a.bbox[1] = d.bbox[1]; a.bbox[2] = d.bbox[2]; a.offset = d.offset;
    a.S = d.S; a.n_obs = d.n_obs; a.cap_planes = d.cap_planes; a.pcap = d.pcap;
    a.n_planes = d.n_planes; a.h_rec = d.h_rec; a.ell_C = d.ell_C; a.ell_d = d.ell_d;
    const size_t lds = dilate_lds_bytes(d.pcap);
    hipError_t e = hipFuncSetAttribute((const void *)k_dilate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_dilate, dim3(d.S), dim3(256), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

} // namespace frx
