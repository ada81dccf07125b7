// hipcc translation unit: the resident round kernel (frx_round_kernel.hpp) and its launcher.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#define FRX_KERNEL_LINKAGE static          // the stage kernels of frx_kernels.hpp belong to frx_device.hip: here only their bodies are used
#include "frx_round_kernel.hpp"

namespace frx {

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
size_t round_args_bytes() { return sizeof(RoundArgs); }
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
    // the arguments by POINTER (k_round<.., ARGP>): copied into the handle's device buffer in front of the launch, on its stream (pageable source: staged before the call returns)
    [[maybe_unused]] auto go_ptr = [&](auto kernel) -> int {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if ((e = hipMemcpyAsync(r.args_dev, &a, sizeof(a), hipMemcpyHostToDevice, (hipStream_t)stream)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)stream, (const RoundArgs *)r.args_dev);
        return (int)hipGetLastError();
    };
    // Measured (profiles/r06_ab_argptr.jsonl, four alternating processes per form on one box, bit-identical plans): 602 -> 301 SGPR spills and 4077 -> 1377 v_readlane
    // reloads buy k_round NOTHING - 25.35 against 25.44 us per round at 32 candidates, 23.05 against 23.60 with one, 24.18 against 24.82 at the stock kappa = 48: an
    // s_load at the use is no faster than a v_readlane there, and neither sits on the leader's dependent chain.  (k_eval_cluster, whose arguments are read once per
    // launch instead of once per round, gains 0.23 us of 17.4: frx_device_eval.hip.)  The by-pointer instantiations are therefore only built with
    // -DFRX_ROUND_ARGPTR_BUILD (and then taken with FRX_ROUND_ARGPTR=1); the default is by value.
#ifdef FRX_ROUND_ARGPTR_BUILD
    static const bool argp = [] { const char *e = std::getenv("FRX_ROUND_ARGPTR"); return e && e[0] == '1'; }();
#endif
    a.rs.cand = r.rs_cand; a.rs.f_last = r.rs_f; a.rs.S = r.rs_S; a.rs.Y = r.rs_Y; a.rs.hs = r.rs_hs; a.rs.newest = r.rs_newest; a.rs.bound = r.rs_bound; a.rs.rinv = r.rs_rinv; a.rs.yy = r.rs_yy; a.rs.vd = r.rs_vd;
    if (r.rs_cand) {                                                   // take-over instantiation: <= 64 pieces, no profile
        if (!n64 || r.prof) return (int)hipErrorInvalidValue;
        return r.E == ROUND_E ? go(k_round<ROUND_E, false, 64, true>) : go(k_round<ROUND_E_SMALL, false, 64, true>);
    }
    if (r.E == ROUND_E) {
        if (r.prof) return n64 ? go(k_round<ROUND_E, true, 64>) : go(k_round<ROUND_E, true, 0>);
#ifdef FRX_ROUND_ARGPTR_BUILD
        if (n64 && argp && r.args_dev) return go_ptr(k_round<ROUND_E, false, 64, false, true>);
#endif
        return n64 ? go(k_round<ROUND_E, false, 64>) : go(k_round<ROUND_E, false, 0>);
    }
    if (r.prof) return n64 ? go(k_round<ROUND_E_SMALL, true, 64>) : go(k_round<ROUND_E_SMALL, true, 0>);
#ifdef FRX_ROUND_ARGPTR_BUILD
    if (n64 && argp && r.args_dev) return go_ptr(k_round<ROUND_E_SMALL, false, 64, false, true>);
#endif
    return n64 ? go(k_round<ROUND_E_SMALL, false, 64>) : go(k_round<ROUND_E_SMALL, false, 0>);
}

} // namespace frx
