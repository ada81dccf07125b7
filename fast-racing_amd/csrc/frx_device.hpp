// Device-side problem description and the host-callable kernel launchers.
// frx_device.hip (hipcc, gfx950) implements the launchers; frx_api.cpp (g++) calls them, so the
// host-side optimiser is built with exactly the flags of the CPU oracle (bit-identical L-BFGS
// arithmetic) while the kernels keep hipcc's device code generation.
#pragma once
#include <cstddef>

#include "frx_math.hpp"

#define FRX_BAND_W 13

namespace frx {

struct DevProblem {
    int B, P, kappa, soft, c2;
    double rho, sumT;
    double inv_kappa;                         // 1 / kappa (quadrature weight, CPU.hpp:259)
    PenaltyConst pc;
    // per candidate
    const int *poff, *coff, *xoff, *boff;     // [B+1] fine pieces, coarse pieces, variables, band offsets (doubles)
    const int *cvoff;                         // [B+1] first waypoint-vertex record of the candidate (vrec is in waypoint order)
    const double *headPVA, *tailPVA;          // [B][9] column-major (p|v|a)
    // per fine piece
    const int *piece_hbeg, *piece_K;          // [P] first half-space record / count of the piece's polytope (idxHs expanded)
    const int *piece_coarse, *piece_iv;       // [P] global coarse index of the piece; interval count of that coarse piece
    // per coarse piece
    const int *coarse_iv, *coarse_fbeg;       // [Pc] interval count, first fine piece (global)
    // per waypoint (candidate b owns N_b-1, global index = poff[b] - b + i)
    const int *wp_vbeg, *wp_nv, *wp_xbeg;     // first vertex record, vertex count, absolute index of its xi segment in x
    const double *hblk;                       // [P][Kmax+1][4]: {origin xyz, K}, then K x (unit normal, n.(p_k - origin) - margin), zero padded
    const double *vrec;                       // waypoint vertices [..][3] in [v0, v_r - v0] form, waypoint order
    // optional (null): command flags of this round per candidate / per fine piece, left by k_lbfgs_pre; candidates without
    // DV_EVAL (finished, or restoring) are skipped by the objective kernels - the tail of a large batch runs at the cost of the
    // candidates still active
    const int *cand_active, *piece_active;
    double *wq_glob;                          // [number of waypoints][4] scratch: {|xi|^2, sum_a V_a xi_a^2} per waypoint, written by the forward map and read by the
                                              // adjoint of the same evaluation (stage kernels; the resident kernel keeps them in LDS, ResidentOps::wq)
    long long *stamps;                        // optional (null): s_memtime stamps of candidate 0's phases, [2][16] (frx_profile_phases); the resident kernel keeps a second copy of one chosen evaluation behind them ([64])
};

enum { SOLVER_KNOT_PCR = 0, SOLVER_BANDED_LU = 1 };

struct LaunchGeom {
    int maxN, maxCN, Kmax, lpp, ppw;       // lpp lanes per piece; ppw = 64 / lpp pieces per WAVE (the resident round kernel's per-wave tasks)
    int pen_w, ppg;                        // stage kernel k_penalty: waves per workgroup (1..4, chosen for lane utilisation) and pieces per workgroup = 64 pen_w / lpp
    size_t lds_fwd, lds_bwd, lds_pen;      // banded-LU kernels + penalty kernel
    size_t lds_pen2;                       // the two-phase form of the penalty kernel (k_penalty_lat2: pen_w = 4, one sample per lane), 0 = not applicable
    int solver;                            // SOLVER_KNOT_PCR (default) | SOLVER_BANDED_LU
    int knot_threads;                      // workgroup size of the knot kernels: 64 * ceil(maxN / 64)
    size_t lds_kfwd, lds_kbwd;
    int maxXb, maxVb;                      // per-candidate maxima: free variables, waypoint-vertex doubles (3 per vertex)
    int pcr_steps;                         // ceil(log2(maxN - 1)): reduction steps whose multipliers k_forward_knot saves
    double *pcrw;                          // [(pcr_steps*8 + 4)][P] saved multipliers + final D^-1 per knot
    int ev_G = 0;                          // one-launch evaluation (frx_eval_kernel.hpp): workgroups per candidate, 0 = not applicable (more than 64 pieces, banded solver)
    size_t lds_ev = 0;
    size_t lds_solo = 0;                   // one workgroup per candidate, one launch per evaluation (frx_solo_kernel.hpp): dynamic LDS, 0 = not applicable
};

// all return a hipError_t value as int (0 = hipSuccess); stream is a hipStream_t
int launch_set_limits(const LaunchGeom &g);
int launch_forward(const DevProblem &dp, const LaunchGeom &g, const double *x, double *T, double *C, double *band, void *stream);
int launch_penalty(const DevProblem &dp, const LaunchGeom &g, const double *T, const double *C, double *out20, void *stream);
int launch_backward(const DevProblem &dp, const LaunchGeom &g, const double *x, const double *T, const double *C,
                    const double *band, const double *out20, double *f, double *grad, void *stream,
                    const double *tap_d = nullptr, const int *tap_flags = nullptr, void *tap_res = nullptr,
                    unsigned *tap_arrive = nullptr, volatile unsigned *tap_flag = nullptr, unsigned tap_round = 0);


// One launch per evaluation (frx_eval_kernel.hpp): clusters of g.ev_G workgroups, one per candidate.  ll: [P][40 + 38] granule words (penalty partials, then (C, T)), words: [64 B + 1] (the last one: status), both
// zeroed ONCE at allocation.  The caller has checked that dp.B * g.ev_G workgroups are resident at once (eval_cluster_geometry).
int eval_cluster_geometry(LaunchGeom &g);                         // fills ev_G / lds_ev from the other fields; returns ev_G
// The handle's constant arguments are packed ONCE (eval_cluster_args into eval_cluster_args_bytes() bytes) and live in host AND device memory; a launch passes the
// device copy by pointer (the by-value form behind FRX_EVAL_ARGPTR=0 reads the host copy) plus what changes per call.
// status_host: one word of mapped host memory that receives the code of an expired wait (the device's own sticky word stays the authority inside the launch).
size_t eval_cluster_args_bytes();
void eval_cluster_args(const DevProblem &dp, const LaunchGeom &g, double *T, double *C, unsigned long long *ll, unsigned *words, void *out);
int launch_eval_cluster(const LaunchGeom &g, int B, const void *args_host, const void *args_dev, const double *x, double *f, double *grad,
                        unsigned long long timeout_ticks, void *stream, unsigned *status_host = nullptr);
// workgroups of k_eval_cluster a CU holds with lds_bytes of dynamic LDS (occupancy query of the runtime; 0 on error)
int eval_cluster_blocks_per_cu(size_t lds_bytes);
int eval_cluster_raise_limit(size_t lds_bytes);                 // raises k_eval_cluster's dynamic-LDS limit on the current device (never lowers it)

// One launch per evaluation for LARGE batches (frx_solo_kernel.hpp): one workgroup per candidate runs forward map, penalty integral and adjoint back to back.
// Same stage buffers, same tap, same results (bit for bit) as launch_forward + launch_penalty + launch_backward.
int eval_solo_geometry(LaunchGeom &g, int samples_per_piece);     // fills lds_solo; returns 1 when the form applies (<= 64 pieces, kappa + 1 <= 64 samples per piece, knot solver)
int eval_solo_raise_limit(const LaunchGeom &g);
int eval_solo_blocks_per_cu(const LaunchGeom &g);                 // workgroups of the kernel a CU holds (occupancy query of the runtime; 0 on error)
int launch_eval_solo(const DevProblem &dp, const LaunchGeom &g, const double *x, double *T, double *C, double *out20, double *f, double *grad, void *stream,
                     const double *tap_d = nullptr, const int *tap_flags = nullptr, void *tap_res = nullptr,
                     unsigned *tap_arrive = nullptr, volatile unsigned *tap_flag = nullptr, unsigned tap_round = 0);

// ---- device-vector L-BFGS (frx_lbfgs_kernels.hpp) ----
struct DvBuffers;
struct DvLaunch {
    const int *xoff; double *x, *g, *xp, *gp, *d, *S, *Y, *ys, *gt; int *dflags = nullptr, *pflags = nullptr; const int *poff = nullptr; size_t ld; size_t hs = 0; int m, B, E, W, PF, BLK;   // k_lbfgs_pre: E doubles per thread, W waves per candidate, PF history rows of look-ahead, BLK pairs per reduction
};
// E doubles per thread x W waves: the smallest padded row 64*W*E that holds n; among equal rows the one with FEWER waves
// (every wave runs the whole serial chain of the recursion and the cross-wave part of a reduction grows with the wave count;
// measured per advance at n = 641: 2 waves x 6 doubles 59.6, 3 x 4 61.6, 6 x 2 65.7 us).
inline void dv_geometry(int n, int *E, int *W, int *PF) {
    size_t best = ~(size_t)0;
    *E = *W = *PF = 0;
    for (int e : {8, 6, 4, 2})
        for (int w = 1; w <= 8; w++) {
            const size_t hs = (size_t)64 * w * e;
            if ((size_t)n <= hs && hs < best) { best = hs; *E = e; *W = w; *PF = e == 2 ? 16 : e == 8 ? 4 : 8; }
        }
}
// Row stride of the history (doubles) for vectors of at most n elements under the geometry (E, W): tight - n + 2 rounded up to 16 - when only a thread's LAST pair can lie
// beyond the row's end (k_lbfgs_pre reads such a pair from the row's zero tail), else the full 64 W E.
inline size_t dv_row_stride(int n, int E, int W, bool tight = true) {
    const size_t HS = (size_t)64 * W * E, hs = ((size_t)n + 2 + 15) & ~(size_t)15;
    if (!tight || hs >= HS) return HS;
    if (HS < 512) return HS;                                            // (short rows measured SLOWER tight: n = 200, 256 candidates, 50.0 -> 54.3 us per advance - profiles/r06_dv_tight_ab.jsonl)
    if (hs / 2 <= (size_t)64 * W * (E / 2 - 1)) return HS;            // (a whole slab of pairs beyond the end: not the case the kernel clamps)
    return hs;
}
int launch_lbfgs_pre(const DvLaunch &dv, const void *cmd, void *res, void *stream);

// ---- resident round kernel (frx_round_kernel.hpp): one launch per plan ----
struct RoundLaunch {
    double *x, *g, *xp, *gp, *d, *f, *T, *C, *out20;              // leader vectors and stage buffers (the handle's own)
    unsigned long long *out20ll = nullptr;                         // [P][20] granules (2 words per value): the penalty partials' way to the adjoint when every candidate has <= 64 pieces; null: out20 + arrival count
    double *pubsyg, *part, *upub, *dpub, *dbg;                     // cluster exchange buffers ([S][3 NXP + 2], [S][G][512], [S][258], [S][NXP] granules)
    unsigned *words;                                               // [ROUND_WORDS_PER_CAND S + 2 + S G + 4 B]: a 512-byte block per cluster (phase, cntA, uflag, cntL in separate lines), then census, status, XCC ids, prediction counters per candidate
    void *h_cmd, *h_res;                                           // mapped host mailboxes, [S] x 16 B (x cmd_stride) and [S] x 64 B
    unsigned long long timeout_ticks;
    unsigned long long *prof = nullptr;                            // optional [B][G][16]: per-segment ticks (profiling instantiation)
    unsigned long long *trace = nullptr; int trace_cap = 0; unsigned trace_lo = 0, trace_hi = 0;   // optional timeline of cluster 0 (profiling instantiation): [G][trace_cap] events of the phases [trace_lo, trace_hi)
    int B, S, G, m, E, NXP;                                         // B candidates on S <= B clusters of G workgroups (S < B: the host hands candidates S .. B-1 to clusters that finish, DV_NEXT)
    int dbg_cap = 0, dbg_cands = 0;                                 // direction log (dbg): [B] counts + dbg_cands x dbg_cap records of 4 NXP + 2 doubles
    double ls_ftol = 1e-4, ls_gtol = 0.9, ls_min_step = 1e-20, ls_max_step = 1e20, ls_xtol = 1e-16;   // frx_lbfgs_params of the plan (leader's prediction of the host's verdict)
    int ls_max_linesearch = 40, speculate = 1;
    int cmd_stride = 4;                                            // h_cmd: cluster k's 16-byte command at 16 * cmd_stride * k
    int fast_control = 1;                                          // see RoundArgs
    int stamp_round = 0;                                           // profiling: keep the cycle stamps of cluster 0's evaluation number stamp_round (0: of its last one)
    // take-over of plans the per-stage rounds began (frx_round_kernel.hpp, RoundArgs::rs): device arrays, [S] each unless stated; rs_cand == nullptr: a plan from its start
    const int *rs_cand = nullptr, *rs_newest = nullptr, *rs_bound = nullptr;
    const double *rs_f = nullptr, *rs_S = nullptr, *rs_Y = nullptr, *rs_rinv = nullptr, *rs_yy = nullptr, *rs_vd = nullptr;   // S, Y: the per-stage history [B][m][rs_hs]; rinv [S][128][129], yy [S][128][128], vd [S][128]
    size_t rs_hs = 0;
    void *args_dev = nullptr;                                       // optional device buffer of round_args_bytes() bytes: the kernel takes its arguments through it (k_round<.., ARGP>)
};
enum { ROUND_E = 56, ROUND_E_SMALL = 28, ROUND_WORDS_PER_CAND = 128 };                                             // history doubles per thread and array of the instantiated kernel
// LDS bytes one workgroup of the round kernel needs (0 = geometry not supported)
size_t round_lds_bytes(const LaunchGeom &g, int m, int E);
size_t round_args_bytes();
int launch_round(const DevProblem &dp, const LaunchGeom &g, const RoundLaunch &r, void *stream);
int launch_lbfgs_post(const DvLaunch &dv, const double *f, const void *cmd, void *res, void *stream);

// ---- corridor cells on the device (frx_corridor_kernels.hpp) ----
struct DilateLaunch {
    const double *p1, *p2, *obs;            // device pointers: [S][3], [S][3], [n_obs][3]
    double bbox[3], offset;
    int S, n_obs, cap_planes, pcap;         // pcap: candidate points a workgroup can hold in LDS
    int *n_planes; double *h_rec, *ell_C, *ell_d;
};
size_t dilate_lds_bytes(int pcap);
int launch_clock_probe(double *out, unsigned long long *stamps, int blocks, unsigned long long ticks, void *stream);   // stamps [2 blocks]: shader cycles, 100 MHz ticks per block
int launch_dilate(const DilateLaunch &d, void *stream);

} // namespace frx
