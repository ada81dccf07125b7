// ONE LAUNCH PER EVALUATION for batches the chip holds at once (frx_objective_eval[_device]: x -> (f, grad f), SURVEY.md 8 row a1;
// the reference's objectiveFunc, se3gcopter_cpu.hpp:961-1000, which its CUDA path splits over a host call, one persistent kernel and a host call again,
// cuda_computer.cu:530-547).
//
// The three stage kernels of an evaluation (forward map, penalty integral, adjoint) are three dependent launches: at the headline batch
// (32 candidates x 64 pieces) each is one wave's dependent chain of 4.6-7.7 us of which 1.5-2.5 us are the launch itself - start of the grid,
// the first trip to HBM for operands the previous kernel just had in LDS, the drain at the end (the same bodies inside the resident round kernel,
// frx_round_kernel.hpp, take 5.2 / 3.1 / 5.2 us).  Here an evaluation is one grid of CLUSTERS, one per candidate:
//   workgroup 0       LEADER: stages x and the polytopes once, runs the forward map, whose (C, T) leave as granules, then - while the members integrate - everything
//                     of the adjoint that does not need the penalty partials, polls the partials, finishes the adjoint, writes f and the gradient
//   workgroups 1..G-1 MEMBERS, every WAVE on its own: fetch the corridor blocks of its wave-task (ppw pieces), wait at the cluster's gate word, poll the task's
//                     (C, T) granules, integrate the penalty, send the 20 partials per piece as granules and leave
// BOTH directions travel as self-validating granules (rk_ll_put, frx_kernels.hpp): no drain, no flag behind the payload, no barrier on either side - the
// round kernel's (C, T) hand-off (write-through stores, drain, barrier, phase word; poll, barrier, loads) measures 1.3 us + a load trip, this one a load trip.
// Same integrand and sums as the stage kernels (f is bit-identical); the adjoint runs in the resident order of operations (backward_knot_wsp64), whose gradient
// differs from the stage kernel's in the last bits (<= 1e-11 relative, tests/test_gpu_parity.py).
//
// Tags instead of zeroed buffers: the launch is captured in hipGraphs (bench.py replays one), so nothing on the host may run between two of them.  Every
// cluster keeps the tag of its last COMPLETED evaluation in device memory (`done`); an evaluation uses tag = done + 1 for its flag and its granules, and the
// leader stores it back as its very last action - behind the arrival of every partial, hence behind every member's read of `done` (a member without a task
// reads nothing and leaves).  Launches on one stream do not overlap, so no workgroup ever sees a tag of the future.
// Every spin is bounded (EvalClusterArgs::timeout_ticks); an expired wait records RK_ERR_PHASE / RK_ERR_ARRIVE in `status` and the candidate's f becomes NaN.
// Residency: a leader waits for members of the SAME launch, so all of a cluster's workgroups have to get a CU.  Blocks are dispatched in index order and a
// cluster's blocks are 8 apart within one group of 8 G consecutive blocks (the XCD mapping of k_round), so the resident blocks of a launch always contain
// whole clusters, which finish and make room; the launcher only takes this path when B G <= the device's CU count (frx_device.hip).
#pragma once
#include <type_traits>

#include "frx_round_kernel.hpp"

namespace frx {

struct EvalClusterArgs {                      // constant for the life of a handle (a copy lives in device memory: ARGP below)
    DevProblem dp;
    double *T, *C;
    ll_u64 *out20ll;                         // [P][20] granules: the penalty partials, members -> leader (its own buffer: the round kernel's carries tags of its own)
    ll_u64 *ctll;                            // [P][19] granules: coefficients and duration of every piece, leader -> members
    unsigned *words;                         // [B][64]: cluster k's 256-byte block: word 0 = gate (tag << 4 | the leader's XCD + 1: (C, T) are out), words 8 .. 8 + G - 2 = tag << 4 | XCD + 1 of
                                             // members 1 .. G-1 (written at their entry), word 32 = tag of the last completed evaluation
    unsigned *status;                        // [1] sticky error word
    int G, maxCN, maxXb, maxVb, nsteps, lpp, ppw, Kmax, pen_lds, maxN19;
};

// LDS of a workgroup (doubles): 2 control | (C, T) copy, x, polytopes, multipliers, waypoint sums, evaluation scratch (leader AND members: both run the forward map) | members: 4 waves x pen_lds
struct EvalClusterLds { int ctl, xs, vs, pw, wq, ev, mem, total; };
__host__ __device__ inline EvalClusterLds eval_cluster_lds(int maxN19, int maxXb, int maxVb, int maxCN, int nsteps, int pen_lds) {
    EvalClusterLds L;
    int o = 2;
    L.ctl = o; o += (maxN19 + 1) & ~1;
    L.xs = o; o += (maxXb + 1) & ~1;
    L.vs = o; o += (maxVb + 1) & ~1;
    L.pw = o; o += ((nsteps * 8 + 5) * 64 + 1) & ~1;
    L.wq = o; o += 4 * 64;
    L.ev = o;
    const int e = 36 * 64 + 9 * 65 + 2 * 64 + maxCN + 16;   // rows | knot arrays | Tf, gT | gCo | cross-wave partials (forward_knot_body / backward_knot_wsp64 with resident operands)
    o += (e + 1) & ~1;
    L.mem = o;                                                          // members (round 6: they run the forward map themselves, in the leader's layout): 4 waves x pen_lds behind it
    L.total = o + 4 * pen_lds + 8;
    return L;
}

// ARGP (round 6): the arguments that are constant for the life of a handle - the problem descriptor and the geometry, ~400 bytes - come
// through a POINTER to a copy in device memory instead of by value, and only what changes per call (x, f, g, the bound, the test switches) stays in the kernarg
// segment.  By value every field is loaded into a scalar register at entry and stays live for the whole kernel: 100 registers' worth of them for a budget of 102 -
// the shipped object had 63 SGPR spills and 767 v_readlane reloads in front of their uses (scripts/isa_report.py).  Behind a pointer a field is an s_load at its
// use (scalar cache) and nothing is kept: 0 spills, 12 v_readlane, 3463 instead of 4093 VALU instructions in the kernel.
struct EvalCallArgs {                         // what changes per call
    const double *x; double *f, *g;
    unsigned *status_host;                   // optional, mapped host memory: the code of an expired wait, where the launcher sees it without a synchronisation (frx_api.cpp: launch_eval)
    rk_u64 timeout_ticks;                    // bound of every spin
    int test_drop_members;                   // tests (frx_debug_set_eval_fused(p, 2)): the members leave at once, as if they never got a CU - the leader's wait for the partials expires
    int force_wt;                            // 1 = every payload store write-through, as if no two workgroups shared an XCD (tests: FRX_EVAL_FUSED_WT=1)
};
template <bool ARGP>
__global__ __launch_bounds__(256, 1) void k_eval_cluster(typename std::conditional<ARGP, const EvalClusterArgs *__restrict__, EvalClusterArgs>::type arg, EvalCallArgs call) {
    const EvalClusterArgs &a = [&]() -> const EvalClusterArgs & { if constexpr (ARGP) return *arg; else return arg; }();
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane8 = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int wg = rest % a.G, k = lane8 + 8 * (rest / a.G);       // (k_round's mapping: a cluster's blocks share blockIdx % 8 - one XCD, as observed)
    if (k >= a.dp.B) return;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = k;
    // (diagnostic, frx_debug_profile_eval_cluster: cycle stamps of cluster 0 - 40..43 the leader: entry, forward map done, adjoint done, end; 44..48 wave 0 of member 1, see penalty_wave_ll - these in ticks of the 100 MHz counter all workgroups share; 49: the leader's shader clock at entry, the origin of the bodies' own stamps 0..31)
    if (a.dp.stamps && k == 0 && wg == 0 && t == 0) { a.dp.stamps[40] = (long long)wall_clock64(); a.dp.stamps[49] = (long long)__builtin_readcyclecounter(); }
    unsigned *flag = a.words + (size_t)k * 64, *done = flag + 32;
    const int p0 = a.dp.poff[c], N = a.dp.poff[c + 1] - p0;
    const int ntasks = (N + a.ppw - 1) / a.ppw;                     // wave-tasks of this candidate: ppw pieces each; members 1 .. G-1 hold 4 (G - 1) >= ntasks waves
    if (wg != 0 && ((wg - 1) * 4 >= ntasks || call.test_drop_members)) return;   // a member without a task
    // A wait of an EARLIER launch expired and nobody has cleared the word yet (the capturable form has no host-synchronous point of its own: replays of a captured
    // graph go on until the caller polls frx_eval_status): nothing is evaluated, nothing spins; the objective values say so.
    if (__builtin_expect(__hip_atomic_load(a.status, FRX_RLX_AGENT) != 0u, 0)) {
        if (wg == 0 && threadIdx.x == 0) call.f[k] = __builtin_nan("");
        return;
    }
    unsigned tag = __hip_atomic_load(done, FRX_RLX_AGENT) + 1u;     // (stays in a vector register: nothing waits for the load until the tag is used)
    if (tag >= (1u << 28)) tag = 1u;
    // Where does this workgroup run?  Payload between two workgroups of one XCD can meet in that XCD's L2 (plain stores, L1-bypassing loads); across XCDs it has
    // to be written through to memory - measured 2.1 us of an 18.5 us evaluation.  The block -> cluster mapping above puts a cluster on one XCD on the hardware
    // this was written on, but nothing relies on it: the members publish their XCD at entry, the leader compares before it sends (C, T), its own XCD travels in
    // the gate word, and whoever finds a partner elsewhere (or not yet heard of) writes through.
    unsigned my_xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc = call.force_wt ? 0u : (my_xcc & 7u) + 1u;                  // 0: "nowhere" - never equal to a partner's
    const EvalClusterLds L = eval_cluster_lds(a.maxN19, a.maxXb, a.maxVb, a.maxCN, a.nsteps, a.pen_lds);
    // Round 6: the MEMBERS run the forward map too - the same body on the same x, so their (C, T) are the leader's bit for bit - instead of waiting for the leader's
    // coefficients: until now (C, T) travelled as granules behind a gate word (sweep 0.7 us, gate seen 0.15, granules polled and staged 0.7 us on the members' side) while
    // the members' four waves had nothing to do for the first 8 us of the launch.  The leader sends nothing; its XCD goes into the gate word at ENTRY (the members look at
    // it when their partials leave, microseconds later; not there yet counts as "elsewhere": write-through).
    double *ctl = sm + L.ctl, *ev = sm + L.ev;
    ResidentOps ro;
    ro.xs = sm + L.xs; ro.vs = sm + L.vs; ro.dsv = sm + L.xs; ro.pw = sm + L.pw; ro.gs = nullptr; ro.vskew = 0; ro.wq = sm + L.wq; ro.gpub = nullptr; ro.gwt = true;
    ro.quiet = wg != 0;
    const int task = (wg - 1) * 4 + wave;                                  // (members) this wave's task: ppw pieces
    const bool has_task = wg != 0 && task < ntasks;
    const int gp0 = p0 + task * a.ppw, npieces = has_task ? min(a.ppw, N - task * a.ppw) : 0;
    const int hstride = (a.Kmax + 1) * 4;
    double *wsm = sm + L.mem + (size_t)wave * a.pen_lds, *hS = wsm, *red = wsm + (size_t)a.ppw * hstride;
    if (wg == 0) { if (t == 0) __hip_atomic_store(flag, (tag << 4) | (my_xcc ? my_xcc : 15u), FRX_RLX_AGENT); }
    else if (has_task) {   // corridor blocks of the wave's task: constant, in flight under the forward map
        if (k == 0 && wg == 1 && wave == 0 && lane == 0 && a.dp.stamps) a.dp.stamps[44] = (long long)wall_clock64();
        const double2 *h2 = (const double2 *)(a.dp.hblk + (size_t)gp0 * hstride);
        const int nh2 = (npieces * hstride) >> 1;
        for (int i0 = lane; i0 < nh2; i0 += 4 * 64) {
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + 64 * u; v[u] = h2[i < nh2 ? i : nh2 - 1]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + 64 * u; if (i < nh2) { hS[2 * i] = v[u].x; hS[2 * i + 1] = v[u].y; } }
        }
    }
    forward_knot_body<false, 64, 5>(a.dp, call.x, a.T, a.C, a.maxCN, a.maxXb, a.maxVb, 64, nullptr, a.nsteps, c, ev, ctl, true, &ro);   // MODE 1 | 4: stages x and the polytopes itself, (C, T) into ctl only
    if (wg != 0) {
        // every WAVE of a member is on its own from here (no workgroup barrier below): its samples out of the workgroup's (C, T) copy, its partials as granules
        if (!has_task) return;
        long long *const mst = (k == 0 && wg == 1 && wave == 0 && lane == 0) ? a.dp.stamps : nullptr;
        if (mst) mst[46] = (long long)wall_clock64();
        const int pl = lane / a.lpp, jl = lane - pl * a.lpp;
        if (pl < npieces) penalty_lane_samples<true>(a.dp, ctl + (size_t)(task * a.ppw + pl) * 19, hS + (size_t)pl * hstride, ctl[(task * a.ppw + pl) * 19 + 18], jl, a.lpp, a.Kmax, red + lane * 21);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (mst) mst[47] = (long long)wall_clock64();
        const unsigned gv = __hip_atomic_load(flag, FRX_RLX_AGENT);           // the leader's XCD, if it has arrived under this evaluation's tag
        const bool wt = my_xcc == 0u || (gv >> 4) != tag || (gv & 15u) != my_xcc;   // plain stores when the leader runs on this XCD (its L2 is the meeting point), write-through otherwise
        penalty_reduce<true>(red, npieces, a.lpp, nullptr, lane, 64, wt, a.out20ll + (size_t)gp0 * 40, tag);
        if (mst) mst[48] = (long long)wall_clock64();
        return;
    }
    // ---- leader ----
    if (a.dp.stamps && k == 0 && t == 0) a.dp.stamps[41] = (long long)wall_clock64();
    ro.o20ll = a.out20ll; ro.o20tag = tag; ro.status = a.status; ro.spin_ticks = call.timeout_ticks;
    const LineSearchTap tap{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, nullptr, nullptr};
    backward_knot_body<true, 64>(a.dp, call.x, a.T, a.C, nullptr, call.f, call.g, a.maxCN, a.maxXb, a.maxVb, 64, nullptr, a.nsteps, tap, c, ev, ctl, &ro);
    if (a.dp.stamps && k == 0 && t == 0) a.dp.stamps[42] = (long long)wall_clock64();
    __syncthreads();
    if (t == 0) {
        const double fv = ev[36 * 64 + 9 * 65 + 2 * 64 + a.maxCN];  // `red[0]` of backward_knot_wsp64: the objective value (a resident caller's f does not go to global memory there)
        const unsigned code = __hip_atomic_load(a.status, FRX_RLX_AGENT);
        const bool bad = code != 0u;
        call.f[c] = bad ? __builtin_nan("") : fv;
        if (bad && call.status_host) __hip_atomic_store(call.status_host, code, FRX_RLX_SYS);
        __hip_atomic_store(done, tag, FRX_RLX_AGENT);
        if (a.dp.stamps && k == 0) a.dp.stamps[43] = (long long)wall_clock64();
    }
}

} // namespace frx
