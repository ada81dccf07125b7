// The resident round kernel: a whole optimisation of B candidates as ONE launch.  (Design: DESIGN.md section 3.5; history of its steps: profiles/NOTES.md.)
//
// What the reference does on its device (cuda_computer.cu:51-64, 384-405, 454-466, 535-548): the penalty kernel is launched once,
// polls a flag in mapped host memory, evaluates, posts `done`, and the host spins on it; every other stage of a round runs on the
// CPU.  Here every stage of a round runs on the device and the kernel stays resident for the whole plan: the host keeps the
// line-search DECISIONS (frx_lbfgs.hpp, SolverDV) and talks to each cluster through a 16-byte command / 64-byte result mailbox
// in mapped host memory; candidates advance independently of each other (no batch-wide round barrier).
//
// Geometry: a CLUSTER of G workgroups (256 threads, one per CU) per candidate, grid = 8 G ceil(S / 8) <= number of CUs; S clusters take
// B >= S candidates one after the other (work queue, DV_NEXT).
//   workgroup 0      LEADER: x, g, xp, gp, d, the waypoint polytopes, the reduction multipliers and (C, T) live in its LDS for the whole plan
//                    (global x is read once and written once); it talks to the host, predicts the host's next command (LineSearch runs on
//                    both sides) and runs the MINCO forward map and the adjoint (forward_knot_body / backward_knot_body of frx_kernels.hpp,
//                    shared arithmetic).  It carries no history: its loop is a separate branch of the kernel, so the evaluation bodies get
//                    the whole register file.
//   workgroups 1..G-2 keep 1/(G-2) of the candidate's (s, y) HISTORY RESIDENT IN REGISTERS: thread (slot j, half h) holds the elements
//                    [h E, (h+1) E) of its workgroup's chunk of s_j and y_j - 2 E doubles (E = 56, or 28 on twice the workgroups when the chip
//                    has room); the history of the headline batch (46 MB) lives in the register files of the chip and is never read from HBM
//                    (k_lbfgs_pre streams it twice per accepted step).  They form s and y themselves from the point and gradient the leader publishes.
//   workgroup G-1    DENSE: the m x m factors of the compact L-BFGS representation, resident in LDS (R^-1) and registers (Y^T Y)
//   workgroups 1..G-2 (and the leader, last in line) evaluate the penalty integrand of their share of the pieces (penalty_body)
//
// Direction: with the history distributed by ELEMENTS, the two-loop recursion (2 m strictly sequential dot products of length n,
// lbfgs.hpp:1381-1411) would need 2 m cross-CU reductions.  The same product  d = -H g  is evaluated in the compact form of
// Byrd-Nocedal-Schnabel (1994, eq. 3.1-3.5 with H0 = gamma I, gamma = y.s / y.y of the newest pair as in lbfgs.hpp:1403):
//       w = R^-1 (S^T g),   v = (D + gamma Y^T Y) w - gamma Y^T g,   u = R^-T v,        R_ij = s_i . y_j (i not newer than j)
//       d = -gamma g - S u + gamma Y w
// i.e. one pass over the history for 4 m dot products (S^T g, Y^T g and the new column S^T y_new, Y^T y_new of R and Y^T Y,
// all from registers), three matrix-vector passes on the dense workgroup, one pass for the linear combination.
// In exact arithmetic this IS the two-loop recursion; in FP64 every accepted step's direction agrees with a host recursion over the same
// pairs to 1e-13 (tests/test_gpu_resident.py, both sizes of E).
//
// Cross-workgroup data: (i) write-through stores + L1-bypassing loads ordered by drained flags / counters (ldg / stg in frx_kernels.hpp;
// MI355X guide, Guideline 16 form R1), (ii) self-validating granules polled by the consumer itself (rk_ll_put below).  Every spin is
// bounded: a wait that expires records an error code in `status`, which ends every workgroup of the launch, and the host driver reports
// the failure or falls back to the one-launch-per-stage path.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "frx_kernels.hpp"
#include "frx_lbfgs_kernels.hpp"

namespace frx {

typedef unsigned long long rk_u64;
// control words of a workgroup in LDS: volatile (re-read after every barrier) AND address-space qualified - through a generic volatile
// pointer every access was a flat_load/flat_store with sc0 sc1 (the slow path to LDS)
typedef volatile __attribute__((address_space(3))) unsigned *rk_ldsword;
#define FRX_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define FRX_RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM

enum { PH_ADV = 1, PH_CT = 2, PH_QUIT = 3, PH_INIT = 4, PH_NEXT = 5 };
enum { RK_OK = 0, RK_ERR_CENSUS = 1, RK_ERR_HOST = 2, RK_ERR_PHASE = 3, RK_ERR_ARRIVE = 4, RK_ERR_DENSE = 5, RK_ERR_UFLAG = 6, RK_ERR_HOST_ABORT = 7, RK_ERR_SPECULATION = 8 };
enum { DV_QUIT = 128, DV_STEP_IS_ONE = 64, DV_NEXT = 32 };   // extra command flags of the resident kernel (frx_lbfgs.hpp: DV_* are < 32); the second one lets
                                               // the leader confirm a predicted command from the command WORD alone (no second read over PCIe); DV_NEXT: this
                                               // candidate is finished, the cluster takes candidate slot | bound << 12 of the batch (work queue, see k_round)

// host -> device: word = seq << 32 | bound << 20 | slot << 8 | flags (written last); step first.  device -> host: seq written last.
struct RoundCmd { rk_u64 word; double step; };                          // cluster k's command is h_cmd[k * cmd_stride] (cmd_stride 4: one cache line per cluster)
struct RoundRes { double f, dg, xx, gg, dginit, pad[2]; rk_u64 seq; };
static_assert(sizeof(RoundRes) == 64, "one result per cache line");

// LDS layout (doubles), shared by the kernel and the host-side size computation
struct RoundLds {
    int ctl, sC, yC, gC, xpC, gpC, zC, pair, role, total;    // offsets; role = eval scratch (leader, members) | dense state (dense workgroup)
    int Rf, vd, va, vb, vc, ve, vw, vv, mv;              // dense state: Rf = R^-1 as [128][129] (row stride 129: conflict-free by row AND by column)
};
enum { RK_RS = 129 };                                    // row stride of Rf
// Control words of a candidate (phase, cntA, uflag, cntL) sit in four different 128-byte lines of ITS OWN 512-byte block.  In round 1
// each kind was a dense [B] array: the phase words of 16 clusters shared one cache line - one L2 channel - that 7 idle workgroups per
// cluster poll every ~100 cycles; at 32 clusters a round took 7 us longer than at one (38.3 vs 31.0 us).
enum { RK_WSTRIDE = 128, RK_WORDS_PER_CAND = 128 };
__host__ __device__ inline RoundLds round_lds(int m, int CHT, int eval_doubles) {
    RoundLds L;
    int o = 0;
    L.ctl = o; o += 48;                                   // 16 unsigned | 8 doubles | 16 profile accumulators
    L.sC = o; o += CHT; L.yC = o; o += CHT; L.gC = o; o += CHT; L.xpC = o; o += CHT; L.gpC = o; o += CHT;
    L.zC = o; o += CHT / 2;                               // E zeros (history workgroups: what the first step of a NEW candidate stores into the slots a finished plan leaves behind)
    L.pair = o; o += 2 * 4 * 128;
    o = (o + 1) & ~1;
    L.role = o;
    int d = o;
    L.Rf = d; d += 128 * RK_RS;
    L.vd = d; d += 128; L.va = d; d += 128; L.vb = d; d += 128; L.vc = d; d += 128; L.ve = d; d += 128; L.vw = d; d += 128; L.vv = d; d += 128;
    L.mv = d; d += 512;                                   // two [2][128] half-sum buffers
    const int e = o + eval_doubles;
    L.total = (d > e ? d : e) + 2;
    (void)m;
    return L;
}

struct RoundArgs {
    DevProblem dp;
    int maxCN, maxXb, maxVb, nrow, nsteps, lpp, ppw, Kmax, pen_lds;   // geometry of the evaluation bodies (LaunchGeom); pen_lds in doubles per wave
    double *x, *g, *xp, *gp, *d, *f, *T, *C, *out20, *pcrw;           // leader-private vectors + the evaluation's stage buffers
    double *pubsyg;          // [S][3 NXP + 2] leader -> cluster: the point x and its gradient g (first 2 NXP doubles, zero beyond n), then (slot, pair count)
    ll_u64 *out20ll;         // [P][20] granules: the penalty partials on their way to the leader's adjoint (use_ll, below); null: plain out20 + arrival count
    double *part;            // [S][G][4][128] cluster -> dense: partial dot products
    double *upub;            // [S][258]      dense -> cluster: -u, gamma w, gamma
    double *dpub;            // [S][NXP] granules      cluster -> leader: direction chunks
    unsigned *phase, *cntA, *uflag, *cntL;   // word k * RK_WSTRIDE of each (k: cluster) (zeroed before every launch); the four bases are 32 words apart
    unsigned *census, *status;               // [1] each (zeroed before every launch)
    unsigned *xcc;                           // [S][G] XCC id + 1 of every workgroup (zeroed before every launch)
    RoundCmd *h_cmd; RoundRes *h_res;        // mapped host memory, [S] each: one mailbox per CLUSTER
    rk_u64 timeout_ticks;                    // bound of every spin, in wall_clock64 ticks (100 MHz)
    rk_u64 census_ticks;                     // bound of the start-up census (all workgroups resident)
    double ls_ftol, ls_gtol, ls_min_step, ls_max_step, ls_xtol;   // line-search constants of the plan (frx_lbfgs_params), for the leader's prediction
    unsigned *spec;                                      // [B][4] counters: rounds started on a predicted ADVANCE / on a predicted trial step, predictions the host's command did not confirm (redone), reserved
    int ls_max_linesearch, speculate;
    int cmd_stride;                                                   // commands are cmd_stride x 16 bytes apart in h_cmd
    int fast_control;                                                 // bit 0 (default on): barrier-free confirmation and the first-trial shortcut of the prediction (off: the long forms, for A/B
                                                                      // measurements); bit 1: PROF experiment, one extra timed read of the command mailbox per round (stamps 19 .. 21)
    int stamp_round;                                                  // PROF: the stamps of cluster 0's evaluation number stamp_round are kept in dp.stamps[32 .. 63]
    int poll_sleep;                                                   // 0..3: s_sleep 1 / 2 / 4 / 8 between polls of phase words and counters, 4: none (FRX_RESIDENT_POLL)
    int maxN19;                                                       // 19 maxN: size of the leader's (C, T) copy
    int B, S, G, m, NXP, eval_doubles, ct_doubles;                    // B candidates, S <= B clusters (S < B: work queue, DV_NEXT);                       // NXP = (G - 2) 2 E: padded vector length (history workgroups x chunk);                       // ct_doubles: leader's LDS copies ((C, T), then x, polytopes, direction, multipliers) at the head of its role region, before the eval scratch
    double *dbg;                             // optional direction log (frx_debug.h, frx_debug_direction_log): [B] record counts, then per candidate c < dbg_cands
    int dbg_cap, dbg_cands;                  // dbg_cap records of 4 NXP + 2 doubles: s, y, g (the pair and the gradient the direction was built from), d, slot, pair count
    rk_u64 *trace; int trace_cap; unsigned trace_lo, trace_hi;      // PROF instantiation only: timeline of cluster 0 - every workgroup's thread 0 appends (segment id << 56 | wall clock) for the
                                                                    // phases trace_lo <= phase number < trace_hi, trace_cap events per workgroup ([G][trace_cap]; FRX_RESIDENT_TRACE)
    rk_u64 *prof;                            // PROF instantiation only: [B][G][16] wall-clock ticks (100 MHz) per segment, see RK_P_*; then [B][16]: histogram of the leaders' waits for a host command
    // RESUME instantiation only (frx_api.cpp: the stragglers of a per-stage batch continue here, mid-plan): cluster k takes candidate cand[k] where the per-stage
    // rounds left it - x, g, xp, gp, d are the handle's own vectors (a.x ... a.d), the history its [B][m][hs] rows in natural element order (k_lbfgs_pre), the
    // dense state (R^-1 by slot with row stride RK_RS, Y^T Y [128][128], D = diag(s.y)) rebuilt by the host from the same rows - the SAME pairs, so the same
    // algorithm: the direction is the one the per-stage path's two-loop recursion would form, to rounding.
    struct Resume { const int *cand; const double *f_last; const double *S, *Y; size_t hs; const int *newest, *bound; const double *rinv, *yy, *vd; } rs;
};
// profile segments (thread 0 of every workgroup accumulates the time since its previous checkpoint into one of these)
enum { RK_P_WAIT_HOST = 0, RK_P_VECTORS = 1, RK_P_FORWARD = 2, RK_P_WAIT_PHASE = 3, RK_P_PASS_A = 4, RK_P_WAIT_PART = 5, RK_P_DENSE_IN = 6, RK_P_SOLVE = 7,
       RK_P_WAIT_U = 8, RK_P_PASS_B = 9, RK_P_PENALTY = 10, RK_P_WAIT_ARRIVE = 11, RK_P_GATHER = 12, RK_P_BACKWARD = 13, RK_P_POST = 14, RK_P_PUBLISH = 15 };

// extra timeline events (no segment sum): ids >= 32
#define RK_TR(id) do { if (PROF && threadIdx.x == 0 && a.trace && v.k == 0 && pseq >= a.trace_lo && pseq < a.trace_hi && tr_n < (unsigned)a.trace_cap) \
                         a.trace[(size_t)v.wg * a.trace_cap + tr_n++] = ((rk_u64)(id) << 56) | ((rk_u64)pseq << 40) | ((rk_u64)wall_clock64() & 0xFFFFFFFFFFull); } while (0)
// scheduling fence between the stages of the dense workgroup's passes (loads on their way | products)
#define RK_CHUNK() __builtin_amdgcn_sched_barrier(0)

// acc += e * (x of lane J of this lane's 16-lane row): the FP64 pipe takes a row broadcast as the DPP operand of a 64-bit FMA (`row_newbcast`, the one
// DPP control it accepts), so a matrix-vector product whose vector lives one element per lane costs ONE instruction per matrix element.  Rounds 2-3
// fetched the element with two v_readlane_b32 into a scalar pair first (rk_bcast): three instructions, the FMA waiting for the scalar write - the
// dense passes ran at 8.5 cycles per instruction.  Every 16-lane row of `x` must hold the same sixteen vector elements.
template <int J> __device__ __forceinline__ void rk_fma_row(double &acc, double x, double e) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(e), "n"(J));
}
template <int... J> __device__ __forceinline__ void rk_dot16(double &acc, double x, const double *e, std::integer_sequence<int, J...>) {
    (rk_fma_row<J>(acc, x, e[J]), ...);
}
template <int... J> __device__ __forceinline__ void rk_dot16x2(double &acc0, double &acc1, double x0, double x1, const double *e, std::integer_sequence<int, J...>) {
    ((rk_fma_row<J>(acc0, x0, e[J]), rk_fma_row<J>(acc1, x1, e[J])), ...);
}
typedef std::make_integer_sequence<int, 16> rk_seq16;
// the four dot products of pass A over a thread's E history elements: element I of the half's chunk sits in lane I % 16 of register I / 16
template <int... I> __device__ __forceinline__ void rk_pass_a_dots(double (&acc)[4], const double *gq, const double *yq, const double *S, const double *Y, std::integer_sequence<int, I...>) {
    ((rk_fma_row<I % 16>(acc[0], gq[I / 16], S[I]), rk_fma_row<I % 16>(acc[1], gq[I / 16], Y[I]),
      rk_fma_row<I % 16>(acc[2], yq[I / 16], S[I]), rk_fma_row<I % 16>(acc[3], yq[I / 16], Y[I])), ...);
}
// A VALU write of `x` must be two wait states old before a DPP operand reads it, and the compiler's hazard recognizer does not see into the asm of
// rk_fma_row.  The wait states are TIED TO THE DATA: every broadcast operand passes through an empty asm in front of the s_nop (so its
// producer - a v_cndmask, a conversion - cannot sink below it) and through another one behind it (so no FMA that reads it can rise above); volatile
// asm statements keep their order.  scripts/check_dpp_hazards.py (tests/test_isa.py) scans the shipped code object for a VALU write of a DPP
// operand less than two wait states in front of its read, whatever a future compiler does with live ranges.
template <int N> __device__ __forceinline__ void rk_dpp_settle(double (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(x[i]));
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(x[i]));
}
template <int N, int M> __device__ __forceinline__ void rk_dpp_settle(double (&x)[N], double (&y)[M]) {
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(x[i]));
#pragma unroll
    for (int i = 0; i < M; i++) asm volatile("" : "+v"(y[i]));
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(x[i]));
#pragma unroll
    for (int i = 0; i < M; i++) asm volatile("" : "+v"(y[i]));
}

// One 16-byte system-scope load of a candidate's command {word, step} from mapped host memory: the two live in one aligned 16-byte
// granule that the host fills step first, word (with the sequence number) last, so a word that carries the expected sequence number
// comes with its step - one PCIe round trip instead of two for the commands whose step is not 1.
__device__ __forceinline__ void rk_load_cmd(const RoundCmd *p, rk_u64 &word, rk_u64 &step) {
    typedef unsigned rk_v4u __attribute__((ext_vector_type(4)));
    rk_v4u r;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    word = (rk_u64)r.x | ((rk_u64)r.y << 32); step = (rk_u64)r.z | ((rk_u64)r.w << 32);
}

// pause between two polls of a word another workgroup will write: a.poll_sleep selects the length (s_sleep takes an immediate)
#define RK_PAUSE(a) do { switch ((a).poll_sleep) { case 0: __builtin_amdgcn_s_sleep(1); break; case 1: __builtin_amdgcn_s_sleep(2); break; \
                                                 case 2: __builtin_amdgcn_s_sleep(4); break; case 3: __builtin_amdgcn_s_sleep(8); break; default: break; } } while (0)

// ---- bounded waits (ONE lane) ----
__device__ __forceinline__ bool rk_expired(const RoundArgs &a, rk_u64 deadline) {
    return __hip_atomic_load(a.status, FRX_RLX_AGENT) != 0u || (rk_u64)wall_clock64() > deadline;
}
__device__ __forceinline__ bool rk_wait_eq(const unsigned *w, unsigned want, const RoundArgs &a) {
    const rk_u64 dl = wall_clock64() + a.timeout_ticks;
    for (unsigned spins = 0;; spins++) {
        if (__hip_atomic_load(w, FRX_RLX_AGENT) == want) return true;
        if ((spins & 31u) == 31u && rk_expired(a, dl)) return false;
        RK_PAUSE(a);
    }
}
__device__ __forceinline__ void rk_fail(const RoundArgs &a, unsigned code) {
    unsigned expect = 0u;
    __hip_atomic_compare_exchange_strong(a.status, &expect, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every storing wave drains its write-through stores, the workgroup meets, then ONE lane publishes (Guideline 16, R1)
__device__ __forceinline__ void rk_drain_and_meet() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---- pieces shared by the two role loops ----
struct RoundView {                       // per-workgroup constants; c, n, xbase, p0, N change when the cluster takes another candidate (DV_NEXT)
    int k, c, wg, t, lane, wave, m, n, xbase, p0, N;
    bool wt;
    double *pub, *part, *upub, *dpub;
};
#define RK_PROF(seg) do { if (PROF && threadIdx.x == 0) { const rk_u64 now_ = wall_clock64(); ((rk_u64 *)(sm + L.ctl + 16))[seg] += now_ - prof_last; prof_last = now_; \
                            if (a.trace && v.k == 0 && pseq >= a.trace_lo && pseq < a.trace_hi && tr_n < (unsigned)a.trace_cap) a.trace[(size_t)v.wg * a.trace_cap + tr_n++] = ((rk_u64)(seg) << 56) | ((rk_u64)pseq << 40) | (now_ & 0xFFFFFFFFFFull); } } while (0)

// penalty share of workgroup `pw` (0..G-2) in a CT phase
template <bool PROF>
__device__ __forceinline__ void rk_penalty_share(const RoundArgs &a, const RoundView &v, double *ev, int pw, unsigned ct_tag) {
    // Wave-tasks go to the history workgroups (pw = 1 .. G-2) first and to the leader (pw = 0) last: at the headline geometry 22 tasks meet 24
    // member waves and the leader takes none - its wave 0 is the one whose thread 0 waits for the PCIe acknowledgements of a deferred result post
    // (~1.5 us on three rounds out of four) and then had its own three pieces still to do, which made it the last wave of the cluster to finish.
    const int npw = a.G - 1, ntasks = (v.N + a.ppw - 1) / a.ppw, per_pass = npw * 4;
    const int rank = (pw == 0 ? npw - 1 : pw - 1);                         // order in which the workgroups take tasks
    for (int base = 0; base < ntasks; base += per_pass) {
        if (base + rank * 4 >= ntasks) break;                               // nothing for this workgroup in this pass (workgroup-uniform: no barrier is skipped by a part of it)
        const int task = base + rank * 4 + v.wave;
        const int np = task < ntasks ? min(a.ppw, v.N - task * a.ppw) : 0;
        penalty_body<true, true>(a.dp, a.T, a.C, a.out20, a.lpp, a.ppw, a.Kmax, v.p0 + task * a.ppw, np, ev + (size_t)v.wave * a.pen_lds, v.lane, v.wt, 64, a.out20ll, ct_tag);   // latency form: a wave has its SIMD to itself
        __syncthreads();
    }
}

// ============================================================================================================================
// LEADER (workgroup 0 of a cluster)
// ============================================================================================================================
template <bool PROF, int NR, bool RESUME>
__device__ __forceinline__ void rk_leader_loop(const RoundArgs &a, RoundView v, const RoundLds &L, double *sm) {
    const int k = v.k, t = v.t, lane = v.lane, wave = v.wave;
    int c = v.c, n = v.n;                                                   // the candidate this cluster works on (changes with DV_NEXT)
    const bool wt = v.wt;
    rk_ldsword ctlU = (rk_ldsword)(unsigned *)(sm + L.ctl);
    double *ctlD = sm + L.ctl + 8, *pair = sm + L.pair, *ctl = sm + L.role, *ev = sm + L.role + a.ct_doubles;
    double *xglob = a.x + v.xbase, *gglob = a.g + v.xbase;                  // global memory: read once at the start, written once at the end
    double *pub = v.pub, *dpub = v.dpub;
    // operands the evaluation bodies find in LDS instead of staging them every call (ResidentOps): at the head of the leader's
    // role region behind the (C, T) copy - x, the waypoint polytopes (constant), the direction, the reduction multipliers
    const int xpad = (a.maxXb + 1) & ~1, vpad = (a.maxVb + a.nrow + 1) & ~1;     // polytopes: one double of skew per waypoint (ResidentOps::vskew)
    ResidentOps ro;
    ro.xs = sm + L.role + ((a.maxN19 + 1) & ~1); ro.vs = ro.xs + xpad; ro.dsv = ro.vs + vpad; ro.pw = ro.dsv + xpad;
    // the leader's vectors live in LDS for the whole plan: x (= ro.xs), g (= ro.gs, written by the adjoint), the previous point and
    // gradient xp / gp and the direction d (= ro.dsv).  Round 1 kept them in global memory "touched by this CU only": every use was a
    // load / store through L2 and the adjoint's gradient stores had to be drained before the round could go on.
    ro.gs = ro.pw + (((a.nsteps * 8 + 5) * a.nrow + 1) & ~1);
    double *x = ro.xs, *g = ro.gs, *xp = ro.gs + xpad, *gp = xp + xpad, *dv = ro.dsv;
    ro.wq = gp + xpad;                                                    // [nrow][4], forward map -> adjoint of the same evaluation
    ro.gpub = pub + a.NXP; ro.gwt = wt;                                   // the adjoint also stores the gradient where the history workgroups read it
    ro.vskew = 1;
    // what the forward map would fetch from the index tables at every call (KnotPre, frx_kernels.hpp), fetched once per candidate.  Plain locals, put into the
    // struct at the call: a struct that lives across the loop is not scalarised (it went to 104 bytes of scratch memory, like round 4's attempt with ResidentOps fields).
    int kp_c0 = 0, kp_cN = 0, kp_cv0 = 0, kp_pc = 0, kp_piv = 1, kp_wnv = 1, kp_wvb = 0, kp_wxb = 0;
    double kp_b0 = 0.0, kp_b1 = 0.0, kp_b2 = 0.0, kp_b3 = 0.0, kp_b4 = 0.0, kp_b5 = 0.0;
    auto load_candidate = [&]() {                                           // candidate c's constants and start point into the leader's LDS (once per plan)
        {
            kp_c0 = a.dp.coff[c]; kp_cN = a.dp.coff[c + 1] - kp_c0; kp_cv0 = a.dp.cvoff[c];
            kp_pc = 0; kp_piv = 1; kp_wnv = 1; kp_wvb = 0; kp_wxb = 0;
            if (t < v.N) { kp_pc = a.dp.piece_coarse[v.p0 + t]; kp_piv = a.dp.piece_iv[v.p0 + t]; }
            const int t2 = t - 64, wq = t2 >= 0 ? (t2 >> 1) : v.N;         // the <= 64-piece form: waypoint of the lane PAIR (t - 64) >> 1
            if (wq < v.N - 1) { const int gw = v.p0 - c + wq; kp_wnv = a.dp.wp_nv[gw]; kp_wvb = a.dp.wp_vbeg[gw]; kp_wxb = a.dp.wp_xbeg[gw]; }
            kp_b0 = kp_b1 = kp_b2 = kp_b3 = kp_b4 = kp_b5 = 0.0;
            if (t2 >= 0 && t2 < 3) {
                kp_b0 = a.dp.headPVA[c * 9 + t2]; kp_b1 = a.dp.headPVA[c * 9 + 3 + t2]; kp_b2 = a.dp.headPVA[c * 9 + 6 + t2];
                kp_b3 = a.dp.tailPVA[c * 9 + t2]; kp_b4 = a.dp.tailPVA[c * 9 + 3 + t2]; kp_b5 = a.dp.tailPVA[c * 9 + 6 + t2];
            }
        }
        const int v0 = a.dp.cvoff[c];
        const double *vsrc = a.dp.vrec + 3 * (size_t)v0;
        for (int w = t >> 2; w < v.N - 1; w += 64) {                       // a quad of lanes copies the polytope of waypoint w, w doubles further on
            const int gw = v.p0 - c + w, beg = 3 * (a.dp.wp_vbeg[gw] - v0), cnt = 3 * a.dp.wp_nv[gw];
            for (int j = t & 3; j < cnt; j += 4) ro.vs[beg + w + j] = vsrc[beg + j];
        }
        for (int i = t; i < n; i += 256) { x[i] = xglob[i]; dv[i] = 0.0; g[i] = 0.0; xp[i] = 0.0; gp[i] = 0.0; }
        __syncthreads();
    };
    load_candidate();
    // An accepted step (lbfgs.hpp:1354-1360: s = x - xp, y = g - gp, then the point becomes the base).  The cluster already HAS the point
    // and its gradient: every trial point goes to the first NXP doubles of `pub` when it is formed and every gradient to the next NXP
    // straight from the adjoint (ResidentOps::gpub), and each history workgroup keeps its chunk of the previous point and gradient in LDS,
    // so it forms s and y itself - bit for bit the values the leader would have sent.  Round 2 published s, y and g here: 2016 stores and
    // their drain (1.5-2 us) between the end of the adjoint and the phase word of every accepted step.
    unsigned nadv_l = 0;                                                    // accepted steps so far (index into the direction log)
    unsigned nadv_tag = 0;                                                  // direction phases of this LAUNCH so far = the tag of the granules the cluster sends back (never reset)
    unsigned nct_tag = 0;                                                   // evaluation phases of this launch so far = the tag of the penalty partials' granules (a.out20ll)
    bool trial_done = false, dg_pending = false;
    // The cluster needs nothing of the leader's own bookkeeping of an accepted step (xp = x, gp = g in its LDS) to start on the new direction -
    // the point and its gradient are in `pub` already - so the phase word leaves first and the copies run while the members work (round 4:
    // the copies sat between the prediction and the phase word, ~1 us of every accepted round in front of the whole direction phase).
    bool accept_pending = false;
    bool adv_published = false;                                             // the ADVANCE phase word of the predicted command went out right behind the prediction (below)
    auto accept_step = [&]() {
        double *row = nullptr;                                              // direction log (tests): the pair and the gradient the direction is built from
        if (__builtin_expect(a.dbg != nullptr, 0) && c < a.dbg_cands && nadv_l < (unsigned)a.dbg_cap) row = a.dbg + a.B + ((size_t)c * a.dbg_cap + nadv_l) * (4 * (size_t)a.NXP + 2);
        for (int i = t; i < n; i += 256) {
            const double xv = x[i], gv = g[i];
            if (row) { row[i] = xv - xp[i]; row[a.NXP + i] = gv - gp[i]; row[2 * a.NXP + i] = gv; }
            xp[i] = xv; gp[i] = gv;
        }
    };
    auto flush = [&](const double *xsrc, const double *gsrc) {              // the plan's result for the host: the point (and its gradient) in global memory
        __syncthreads();
        for (int i = t; i < n; i += 256) { xglob[i] = xsrc[i]; gglob[i] = gsrc[i]; }
    };
    rk_u64 prof_last = PROF ? wall_clock64() : 0;
    unsigned tr_n = 0; (void)tr_n;
    unsigned pseq = 0, nphase = 0;
    rk_u64 hseq = 0;
    int lstage = 0, flags = 0, jnew = 0, bound = 0;
    double step = 0.0;
    // Optimistic acceptance.  The host's verdict on the FIRST trial of a line search is a two-line test on numbers the leader already
    // holds (lbfgs.hpp:835-850: sufficient decrease and curvature), and an accepted step is always followed by the same command:
    // ADVANCE | TRIAL | EVAL with the next history slot and step 1 (lbfgs.hpp:1418).  When the test holds the leader starts that command
    // at once instead of idling for the mailbox round trip (3-9 us of a ~55 us round, measured) and CONFIRMS it against the command the
    // host actually sent before the next result is posted.  The host keeps every decision: if it stopped instead (convergence,
    // iteration limit) the leader restores the accepted point and leaves; any other disagreement ends the launch with
    // RK_ERR_SPECULATION and the plan is re-run on the per-stage path.
    bool unconfirmed = false, spec_ready = false, have_cmd = false, ls_ok = false;
    unsigned n_pred_adv = 0, n_pred_trial = 0, n_redone = 0;                // counters of a.spec, written once at the end
    int pred_kind = 0, run_kind = 0;                                        // 1 = ADVANCE predicted, 2 = another trial of the running search predicted (kind of the NEXT / of the RUNNING round)
    double pred_step = 1.0;
    LineSearch ls;                                                          // the host's More-Thuente state machine (frx_lbfgs.hpp), run in step with it on the same numbers
    frx_lbfgs_params lpm = {};                                              // only the line-search constants are consulted on the device
    lpm.f_dec_coeff = a.ls_ftol; lpm.s_curv_coeff = a.ls_gtol; lpm.min_step = a.ls_min_step; lpm.max_step = a.ls_max_step; lpm.xtol = a.ls_xtol; lpm.max_linesearch = a.ls_max_linesearch;
    rk_u64 pred_word = 0, seq_pending = 0;
    double f_acc = 0.0, gg0 = 0.0;
    int last_slot = -1, last_bound = 0;
    if (RESUME) {
        // Take-over of a plan the per-stage rounds began (RoundArgs::rs): the leader's vectors are the per-stage path's own buffers, the last objective value
        // comes with the launch, and ONE phase INIT hands the cluster its state - the previous point and gradient (from `pub`, as in every INIT), the history
        // rows into the members' registers, R^-1, Y^T Y and D into the dense workgroup.  Then `pub` takes the current point and its gradient: what the members
        // form the next pair from.  The host's first command is the plan's PENDING one (a trial of a running search, an ADVANCE, a RESTORE).
        const double *gsrc = a.g + v.xbase, *xps = a.xp + v.xbase, *gps = a.gp + v.xbase, *dsrc = a.d + v.xbase;
        for (int i = t; i < n; i += 256) {
            const double xpv = xps[i], gpv = gps[i];
            g[i] = gsrc[i]; xp[i] = xpv; gp[i] = gpv; dv[i] = dsrc[i];
            stg<true>(pub + i, xpv, wt); stg<true>(pub + a.NXP + i, gpv, wt);
        }
        if (t == 0) ctlD[0] = a.rs.f_last[k];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rk_drain_and_meet();
        pseq++;
        if (t == 0) __hip_atomic_store(a.phase + k * RK_WSTRIDE, (pseq << 4) | (unsigned)PH_INIT, FRX_RLX_AGENT);
        rk_drain_and_meet();
        if (t == 0) __hip_atomic_fetch_add(a.cntL + k * RK_WSTRIDE, 1u, FRX_RLX_AGENT);
        nphase++;
        if (t == 0) { const bool okw = rk_wait_eq(a.cntL + k * RK_WSTRIDE, (unsigned)a.G * nphase, a); if (!okw) rk_fail(a, RK_ERR_ARRIVE); ctlU[0] = okw ? 1u : 0u; }
        __syncthreads();
        const bool ok0 = ctlU[0] != 0u;
        __syncthreads();
        if (!ok0) {
            pseq++;
            if (t == 0) { __hip_atomic_store(&a.h_res[k].seq, ~(rk_u64)0, FRX_RLX_SYS); __hip_atomic_store(a.phase + k * RK_WSTRIDE, (pseq << 4) | (unsigned)PH_QUIT, FRX_RLX_AGENT); }
            return;
        }
        for (int i = t; i < n; i += 256) { stg<true>(pub + i, x[i], wt); stg<true>(pub + a.NXP + i, g[i], wt); }   // (drained in front of the next phase word, like every publication)
    }
    for (;;) {
        int kind = 0;
        RK_TR(40);                                                          // loop top
        if (__builtin_expect(lstage == 0 && spec_ready, 1)) {               // the predicted command, unconfirmed for now (the common case: code laid out for it)
            spec_ready = false; unconfirmed = true; run_kind = pred_kind;
            hseq++;
            flags = (int)(pred_word & 0xFFu) & ~(int)DV_STEP_IS_ONE;
            if (pred_kind == 1) n_pred_adv++; else n_pred_trial++;           // (registers: a read-modify-write of a.spec here cost the leader's first wave a trip to memory per round)
            if (pred_kind == 1) {
                jnew = (int)((pred_word >> 8) & 0xFFFu); bound = (int)((pred_word >> 20) & 0xFFFu);
                step = 1.0;
                f_acc = ctlD[0];
                last_slot = jnew; last_bound = bound;
                if (!adv_published && t == 0) { stg<true>(pub + 2 * a.NXP, (double)jnew, wt); stg<true>(pub + 2 * a.NXP + 1, (double)bound, wt); }
                kind = PH_ADV; accept_pending = true;                       // xp = x, gp = g: behind the phase word (same as the DV_ADVANCE branch below)
            } else step = pred_step;                                        // another trial of the running search: straight to x = xp + step d
            lstage = 1;
        } else if (lstage == 0) {
            if (!have_cmd) {
            if (t == 0) {
                const rk_u64 tw0 = wall_clock64(), dl = tw0 + a.timeout_ticks;
                rk_u64 w = 0, stp = 0;
                bool ok = true;
                for (unsigned spins = 0;; spins++) {
                    rk_load_cmd(a.h_cmd + k * a.cmd_stride, w, stp);
                    if ((w >> 32) == hseq + 1) break;
                    if ((spins & 15u) == 15u && rk_expired(a, dl)) { ok = false; break; }
                }
                if (PROF && a.prof) {                                       // histogram of the waits for a command: bin k = shorter than 2^k us (behind the [B][G][16] segment sums)
                    const unsigned us = (unsigned)((wall_clock64() - tw0) / 100);
                    int bin = 0;
                    while (bin < 15 && (1u << bin) <= us) bin++;
                    a.prof[(size_t)a.S * a.G * 16 + (size_t)k * 16 + bin] += 1;
                }
                if (!ok) { rk_fail(a, RK_ERR_HOST); w = DV_QUIT; __hip_atomic_store(&a.h_res[k].seq, ~(rk_u64)0, FRX_RLX_SYS); }
                ctlU[1] = (unsigned)w;
                if (ok) ctlD[5] = __longlong_as_double((long long)stp);
            }
            __syncthreads();
            hseq++;
            }
            have_cmd = false;                                               // (a command the confirmation step fetched after a wrong prediction is decoded like any other)
            RK_PROF(RK_P_WAIT_HOST);
            const unsigned w = ctlU[1];
            flags = (int)(w & 0xFFu) & ~(int)DV_STEP_IS_ONE; jnew = (int)((w >> 8) & 0xFFFu); bound = (int)((w >> 20) & 0xFFFu);
            step = ctlD[5];
            __syncthreads();
            if (flags & DV_QUIT) { kind = PH_QUIT; flush(x, g); }
            else if (flags & DV_NEXT) {
                // Work queue (more candidates than clusters): the host has finished candidate c - its point goes to global memory as with
                // QUIT - and hands this cluster candidate jnew | bound << 12.  The cluster stays on the chip: the leader loads the new
                // candidate's polytopes and start point, the members re-bind their penalty share, the dense workgroup clears R^-1 and Y^T Y
                // (phase NEXT); the history registers need nothing - pairs are valid by (slot, pair count) of the new plan's steps.
                flush(x, g);
                if (t == 0 && a.spec) { a.spec[c * 4] = n_pred_adv; a.spec[c * 4 + 1] = n_pred_trial; a.spec[c * 4 + 2] = n_redone; }
                n_pred_adv = n_pred_trial = n_redone = 0;
                __syncthreads();
                const int n_old = n;
                c = jnew | (bound << 12);
                v.c = c; v.xbase = a.dp.xoff[c]; n = v.n = a.dp.xoff[c + 1] - v.xbase; v.p0 = a.dp.poff[c]; v.N = a.dp.poff[c + 1] - v.p0;
                xglob = a.x + v.xbase; gglob = a.g + v.xbase;
                load_candidate();
                for (int i = n + t; i < n_old; i += 256) { stg<true>(pub + i, 0.0, wt); stg<true>(pub + a.NXP + i, 0.0, wt); }   // s and y are zero beyond n (a shorter candidate after a longer one)
                if (t == 0) stg<true>(pub + 2 * a.NXP, (double)c, wt);
                nadv_l = 0; trial_done = false; dg_pending = false; spec_ready = false; unconfirmed = false; ls_ok = false; pred_kind = 0;
                last_slot = -1; last_bound = 0;
                kind = PH_NEXT;
            }
            else if (flags & DV_RESTORE) {                                  // lbfgs.hpp:1287-1288; no evaluation follows
                for (int i = t; i < n; i += 256) { x[i] = xp[i]; g[i] = gp[i]; }
                rk_drain_and_meet();
                if (t == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); __hip_atomic_store(&a.h_res[k].seq, hseq, FRX_RLX_SYS); }
                continue;
            } else if (flags & DV_ADVANCE) {                                // lbfgs.hpp:1354-1360: s = x - xp, y = g - gp; the point becomes the base
                f_acc = ctlD[0]; last_slot = jnew; last_bound = bound;
                if (t == 0) { stg<true>(pub + 2 * a.NXP, (double)jnew, wt); stg<true>(pub + 2 * a.NXP + 1, (double)bound, wt); }   // the step's slot and pair count ride along
                kind = PH_ADV; lstage = 1; accept_pending = true;
            } else {
                if (flags & DV_INIT) {                                      // d = -g, xp = x, gp = g (lbfgs.hpp:1220, 1262-1263)
                    f_acc = ctlD[0]; gg0 = ctlD[3]; last_slot = -1; last_bound = 0;
                    // the history workgroups take their chunks of the start point and its gradient as "previous point" (phase INIT, once
                    // per plan): the gradient is in `pub` already (adjoint), the start point is not - it was never a trial point
                    for (int i = t; i < n; i += 256) { const double gv = g[i], xv = x[i]; dv[i] = -gv; xp[i] = xv; gp[i] = gv; stg<true>(pub + i, xv, wt); }
                    if (t == 0) { stg<true>(pub + 2 * a.NXP, 0.0, wt); stg<true>(pub + 2 * a.NXP + 1, 1.0, wt); }   // the plan's first accepted step: slot 0, one pair (see the gather)
                    kind = PH_INIT;
                }
                lstage = 1;
            }
        }
        // One iteration = one COMMAND: its own phase (ADVANCE / INIT / NEXT / QUIT), then - pass 1 - its evaluation, as straight-line code (the two
        // passes are unrolled).  Until round 4 every PHASE was a trip of the outer loop: measured with the timeline (scripts/r04/round_timeline.py),
        // the trip back to the loop top cost 0.6-0.9 us each time - register shuffles and reloads of spilled scalars at the latch - twice per round on the critical path.
        bool leave = false;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
        if (lstage == 1 && kind == 0) {
            if ((flags & DV_TRIAL) && !trial_done) {                        // x = xp + step * d (lbfgs.hpp:825-826); after an ADVANCE the gather above has done it
                __syncthreads();
                for (int i = t; i < n; i += 256) { const double xv = xp[i] + step * dv[i]; x[i] = xv; stg<true>(pub + i, xv, wt); }   // (drained with the forward map's stores before the CT phase word)
            }
            trial_done = false;
            if (flags & DV_EVAL) {
                RK_TR(41);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // x is complete in LDS (no vmcnt: the trial point's stores to `pub` drain behind the forward map)
                RK_PROF(RK_P_VECTORS);
                const KnotPre kpre{v.p0, v.N, kp_c0, kp_cN, v.xbase, kp_cv0, kp_pc, kp_piv, kp_wnv, kp_wvb, kp_wxb, {kp_b0, kp_b1, kp_b2, kp_b3, kp_b4, kp_b5}};
                forward_knot_body<true, NR>(a.dp, a.x, a.T, a.C, a.maxCN, a.maxXb, a.maxVb, a.nrow, a.pcrw, a.nsteps, c, ev, ctl, wt, &ro, NR == 64 ? &kpre : nullptr);
                kind = PH_CT; lstage = 2;
                RK_PROF(RK_P_FORWARD);
            } else {
                rk_drain_and_meet();
                if (t == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); __hip_atomic_store(&a.h_res[k].seq, hseq, FRX_RLX_SYS); }
                lstage = 0;
                break;
            }
        }
        if (PROF && a.dp.stamps && k == 0 && t == 0 && kind == PH_CT) a.dp.stamps[13] = (long long)__builtin_readcyclecounter();
        RK_PROF(RK_P_DENSE_IN);                                             // (leader: command decoded / accepted step taken over)
        if (adv_published && kind == PH_ADV) adv_published = false;         // this phase word is out already (behind the prediction, below)
        else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (PROF && a.dp.stamps && k == 0 && t == 0 && kind == PH_CT) a.dp.stamps[14] = (long long)__builtin_readcyclecounter();
        rk_drain_and_meet();                                                // everything published so far has left this CU
        RK_PROF(RK_P_SOLVE);                                                // (leader: drain of this phase's publications)
        pseq++;
        if (t == 0) __hip_atomic_store(a.phase + k * RK_WSTRIDE, (pseq << 4) | (unsigned)kind, FRX_RLX_AGENT);
        if (PROF && a.dp.stamps && k == 0 && t == 0 && kind == PH_CT) a.dp.stamps[15] = (long long)__builtin_readcyclecounter();
        }
        if (seq_pending != 0) {
            // The result of the round whose acceptance the leader predicted goes to the host only NOW, behind the phase word of the step it
            // started: posted right after the adjoint (round 2), its five stores to host memory sat in front of this publication's drain -
            // ~1.5 us of PCIe write acknowledgements between the end of the adjoint and the cluster's start on the new direction, on three
            // rounds out of four.  Thread 0 alone waits for them (the others meet it at the arrival barrier, behind which the whole
            // direction phase lies anyway); no release fence: its L2 write-back would serve cached stores, and there are none to publish.
            if (t == 0) {
                RoundRes *r = a.h_res + k;
                __hip_atomic_store((rk_u64 *)&r->f, (rk_u64)__double_as_longlong(ctlD[0]), FRX_RLX_SYS);
                __hip_atomic_store((rk_u64 *)&r->dg, (rk_u64)__double_as_longlong(ctlD[1]), FRX_RLX_SYS);
                __hip_atomic_store((rk_u64 *)&r->xx, (rk_u64)__double_as_longlong(ctlD[2]), FRX_RLX_SYS);
                __hip_atomic_store((rk_u64 *)&r->gg, (rk_u64)__double_as_longlong(ctlD[3]), FRX_RLX_SYS);
                __hip_atomic_store((rk_u64 *)&r->dginit, (rk_u64)__double_as_longlong(ctlD[4]), FRX_RLX_SYS);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&r->seq, seq_pending, FRX_RLX_SYS);
            }
            seq_pending = 0;
        }
        if (accept_pending) { accept_step(); accept_pending = false; }      // (the gather below reads gp behind the arrival barrier)
        RK_PROF(RK_P_PUBLISH);
        if (__builtin_expect(kind == PH_QUIT, 0)) { leave = true; break; }
        if (kind == PH_CT) { nct_tag++; rk_penalty_share<PROF>(a, v, ev, 0, nct_tag); RK_PROF(RK_P_PENALTY); }
        // ---- the phase is complete when every workgroup of the cluster has reported ----
        rk_drain_and_meet();
        if (t == 0) __hip_atomic_fetch_add(a.cntL + k * RK_WSTRIDE, 1u, FRX_RLX_AGENT);
        RK_TR(37);
        nphase++;
        // (A direction phase is not waited for here: the direction arrives as granules that the gather below polls, and the arrivals of this phase are
        // part of the count the next phase that DOES wait asks for.  The same holds for an evaluation phase whose penalty partials travel as granules
        // - a.out20ll, the <= 64-piece geometry: the adjoint polls them itself.)
        bool ok = true;
        if (kind != PH_ADV && !(kind == PH_CT && a.out20ll != nullptr)) {
            if (t == 0) { const bool okw = rk_wait_eq(a.cntL + k * RK_WSTRIDE, (unsigned)a.G * nphase, a); if (!okw) rk_fail(a, RK_ERR_ARRIVE); ctlU[0] = okw ? 1u : 0u; }
            __syncthreads();
            ok = ctlU[0] != 0u;
            __syncthreads();
        }
        RK_PROF(RK_P_WAIT_ARRIVE);
        if (__builtin_expect(!ok, 0)) {                                     // tell the host and the cluster, then leave
            flush(x, g);
            pseq++;
            if (t == 0) { __hip_atomic_store(&a.h_res[k].seq, ~(rk_u64)0, FRX_RLX_SYS); __hip_atomic_store(a.phase + k * RK_WSTRIDE, (pseq << 4) | (unsigned)PH_QUIT, FRX_RLX_AGENT); }
            leave = true; break;
        }
        if (__builtin_expect(kind == PH_NEXT, 0)) {                         // every workgroup of the cluster is on the new candidate: tell the host, whose next command starts its plan
            if (t == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); __hip_atomic_store(&a.h_res[k].seq, hseq, FRX_RLX_SYS); }
            lstage = 0;
            break;
        }
        if (kind == PH_ADV) {                                               // gather the direction; dginit = gp . d (lbfgs.hpp:756)
            // ... and, in the same sweep, the first trial point of the new search x = xp + step d (lbfgs.hpp:825-826; every ADVANCE command
            // carries TRIAL): the element a thread gathers is the element it moves, so no barrier lies between the two
            double acc = 0.0;
            nadv_tag++;
            const bool with_trial = (flags & DV_TRIAL) != 0;
            for (int i0 = t; i0 < n; i0 += 3 * 256) {                       // three elements per thread and trip, their loads in flight together (the plain loop
                double dl[3];                                               // compiled to load / s_waitcnt vmcnt(0) / use per element: three L2 round trips in a row)
                {   // the elements travel as granules tagged with the number of the direction phase (rk_ll_put by the history workgroups): poll until all three are this phase's
                    rk_u64 w[3][2];
                    const rk_u64 dl_t = wall_clock64() + a.timeout_ticks;
                    for (unsigned spins = 0;; spins++) {
#pragma unroll
                        for (int u = 0; u < 3; u++) {
                            const int i = i0 + 256 * u;
                            const rk_u64 *gsl = (const rk_u64 *)dpub + 2 * (size_t)(i < n ? i : n - 1);
                            w[u][0] = __hip_atomic_load(gsl, FRX_RLX_AGENT); w[u][1] = __hip_atomic_load(gsl + 1, FRX_RLX_AGENT);
                        }
                        if (rk_ll_ok(w[0][0], w[0][1], nadv_tag) && rk_ll_ok(w[1][0], w[1][1], nadv_tag) && rk_ll_ok(w[2][0], w[2][1], nadv_tag)) break;
                        if ((spins & 31u) == 31u && rk_expired(a, dl_t)) { rk_fail(a, RK_ERR_ARRIVE); break; }
                        RK_PAUSE(a);
                    }
#pragma unroll
                    for (int u = 0; u < 3; u++) dl[u] = rk_ll_value(w[u][0], w[u][1]);
                }
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const int i = i0 + 256 * u;
                    if (i < n) {
                        const double di = dl[u];
                        dv[i] = di; acc += gp[i] * di;
                        if (with_trial) { const double xv = xp[i] + step * di; x[i] = xv; stg<true>(pub + i, xv, wt); }
                    }
                }
            }
            trial_done = with_trial;
            // The NEXT accepted step's slot and pair count follow from this one's, whenever that step comes: they go to `pub` HERE - every workgroup of the
            // cluster has read this step's (the direction is back) - and drain with the forward map's publication.  Until round 4 they were stored behind the
            // prediction, and the phase word of every predicted ADVANCE waited for their acknowledgement from L2 (0.3-0.4 us of every accepted round).
            if (t == 0) {
                const int nslot = jnew + 1 == v.m ? 0 : jnew + 1, nbound = min(v.m, bound + 1);
                stg<true>(pub + 2 * a.NXP, (double)nslot, wt); stg<true>(pub + 2 * a.NXP + 1, (double)nbound, wt);
            }
            if (__builtin_expect(a.dbg != nullptr, 0) && c < a.dbg_cands && nadv_l < (unsigned)a.dbg_cap) {     // direction log (tests): what came back for the pair logged by accept_step
                const size_t rec = 4 * (size_t)a.NXP + 2;
                double *row = a.dbg + a.B + ((size_t)c * a.dbg_cap + nadv_l) * rec;
                for (int i = t; i < n; i += 256) row[3 * a.NXP + i] = dv[i];
                if (t == 0) { row[4 * a.NXP] = (double)last_slot; row[4 * a.NXP + 1] = (double)last_bound; a.dbg[c] = (double)(nadv_l + 1); }
            }
            nadv_l++;
            const double ws = wave_sum_dpp(acc);
            if (lane == 0) pair[wave] = ws;                                 // summed by thread 0 in front of the adjoint: the barriers of the forward map lie in between
            dg_pending = true;
            RK_PROF(RK_P_GATHER);
        }
        if (lstage == 2) {                                                  // after the penalty phase: adjoint, gradient, line-search scalars
            if (dg_pending) { if (t == 0) ctlD[4] = (pair[0] + pair[1]) + (pair[2] + pair[3]); dg_pending = false; }   // gp . d of this round's ADVANCE (read behind the adjoint's barrier)
            LineSearchTap tap{a.d, nullptr, nullptr, nullptr, nullptr, 0u, ctlD};
            if (unconfirmed && !(a.fast_control & 4)) tap.early_cmd = &a.h_cmd[k * a.cmd_stride].word;   // the host's command for a predicted round: thread 0 reads it while the adjoint runs (ctlD[7], ctlD[6])
            ro.o20ll = a.out20ll; ro.o20tag = nct_tag; ro.status = a.status; ro.spin_ticks = a.timeout_ticks;
            backward_knot_body<true, NR>(a.dp, a.x, a.T, a.C, a.out20, a.f, a.g, a.maxCN, a.maxXb, a.maxVb, a.nrow, a.pcrw, a.nsteps, tap, c, ev, ctl, &ro);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // gradient and line-search sums are in LDS; the gradient's copy in `pub` drains before the next phase word
            RK_PROF(RK_P_BACKWARD);
            if (PROF && a.dp.stamps && k == 0 && t == 0) a.dp.stamps[18] = (long long)__builtin_readcyclecounter();   // (18 .. 21: the leader's work behind the adjoint)
            if (PROF && a.dp.stamps && (a.fast_control & 2) && t == 0) {       // experiment (FRX_RESIDENT_TIMED_READ=1 + profile): one extra, TIMED read of the command mailbox per round
                const rk_u64 q0 = wall_clock64();
                rk_u64 w_ = 0, s_ = 0;
                rk_load_cmd(a.h_cmd + k * a.cmd_stride, w_, s_);
                const rk_u64 dq = wall_clock64() - q0;
                if (k == 0) { a.dp.stamps[19] += (long long)dq; a.dp.stamps[20] += 1; if ((long long)dq > a.dp.stamps[21]) a.dp.stamps[21] = (long long)dq; }
            }
            if (unconfirmed) {
                // Common case first, without a barrier: the word thread 0 read while the adjoint ran (ctlD[7], in LDS behind the adjoint's last
                // barrier) is this round's command and equals the predicted one - every thread sees the same word and decides alike.
                // A predicted TRIAL step is confirmed on the step's own 64 bits, read with the word (the word carries only a 24-bit fold of them).
                const rk_u64 we = (rk_u64)__double_as_longlong(ctlD[7]), se = (rk_u64)__double_as_longlong(ctlD[6]);
                const bool step_ok = run_kind != 2 || se == (rk_u64)__double_as_longlong(step);
                if ((a.fast_control & 1) && !(a.fast_control & 4) && (we >> 32) == hseq && (unsigned)we == (unsigned)pred_word && step_ok) unconfirmed = false;
            }
            if (__builtin_expect(unconfirmed, 0)) {                         // the command this round ran on: did the host really send it?
                if (t == 0) {
                    const rk_u64 dl = wall_clock64() + a.timeout_ticks;
                    rk_u64 w = 0, stp_now = 0;                                  // word and step in ONE 16-byte read: they belong together
                    bool ok = true;
                    for (unsigned spins = 0;; spins++) {
                        rk_load_cmd(a.h_cmd + k * a.cmd_stride, w, stp_now);
                        if ((w >> 32) == hseq) break;
                        if ((spins & 15u) == 15u && rk_expired(a, dl)) { ok = false; break; }
                    }
                    // 0 confirmed; 1 the host stopped (QUIT): the accepted point is the result; 2 a disagreement that cannot be undone (the history has
                    // moved on, or the accepted point was overwritten): the launch ends and the plan is re-run per stage; 3 a predicted TRIAL step the
                    // host did not send - it sent another command of the same search (a different step, RESTORE): the evaluation is thrown away and the
                    // host's command executed (xp, gp, d are untouched by a trial)
                    unsigned verdict = 0u;
                    if (!ok) { rk_fail(a, RK_ERR_HOST); verdict = 2u; }
                    else if ((unsigned)w != (unsigned)pred_word || (run_kind == 2 && stp_now != (rk_u64)__double_as_longlong(step))) {
                        if ((unsigned)w & (unsigned)(DV_QUIT | DV_NEXT)) { verdict = 1u; ctlU[2] = (unsigned)w; }
                        else verdict = (run_kind == 2 && !((unsigned)w & (unsigned)(DV_ADVANCE | DV_INIT))) ? 3u : 2u;
                    }
                    if (verdict == 2u) { rk_fail(a, RK_ERR_SPECULATION); __hip_atomic_store(&a.h_res[k].seq, ~(rk_u64)0, FRX_RLX_SYS); }
                    if (verdict == 3u) { ctlU[2] = (unsigned)w; ctlD[5] = __longlong_as_double((long long)stp_now); n_redone++; }   // the whole command came with the read above
                    ctlU[1] = verdict;
                }
                __syncthreads();
                const unsigned verdict = ctlU[1];
                __syncthreads();
                unconfirmed = false;
                if (verdict == 3u) {
                    if (t == 0) ctlU[1] = ctlU[2];                          // where the command decoder looks
                    have_cmd = true; ls_ok = false; lstage = 0;             // no more predictions in this search: the device's copy of the line search has left the host's
                    __syncthreads();
                    break;
                }
                if (verdict == 1u && (ctlU[2] & (unsigned)DV_NEXT)) {           // the host stopped this plan and hands over the next candidate: the accepted
                    for (int i = t; i < n; i += 256) { x[i] = xp[i]; g[i] = gp[i]; }   // point is the result (the command decoder flushes x, g)
                    if (t == 0) ctlU[1] = ctlU[2];
                    have_cmd = true; ls_ok = false; lstage = 0;
                    __syncthreads();
                    break;
                }
                if (verdict != 0u) {
                    if (verdict == 1u) flush(xp, gp); else flush(x, g);                     // host stopped: the accepted point is the result
                    rk_drain_and_meet();
                    pseq++;
                    if (t == 0) __hip_atomic_store(a.phase + k * RK_WSTRIDE, (pseq << 4) | (unsigned)PH_QUIT, FRX_RLX_AGENT);
                    leave = true; break;
                }
            }
            RK_PROF(RK_P_PASS_A);                                           // (leader: confirmation of the command this round ran on)
            // Prediction of the host's next command.  The host feeds (f, g.d) of this trial to its More-Thuente search (SolverDV::feed ->
            // LineSearch::mt_begin / mt_feed, lbfgs.hpp:743-935) and answers with either ADVANCE | TRIAL | EVAL, next slot, step 1 (the trial is
            // accepted: lbfgs.hpp:1418) or TRIAL | EVAL with the search's next step.  Every thread of the leader runs the SAME search object on the
            // same numbers (same source, no contraction on either side), so the answer is known here ~10 us before it could arrive over PCIe
            // (leader's wait for a command at 32 candidates: 8-16 us, profiles/r03_hostwait_probe.jsonl).  The leader starts on it at once and
            // checks it against the host's command word before the NEXT result is posted; the host keeps every decision.  Round 2 predicted only
            // the acceptance of a search's first trial.
            pred_kind = 0;
            if (a.speculate) {
                const double fv = ctlD[0], dgv = ctlD[1];
                bool accepted_first = false;
                if (flags & (DV_ADVANCE | DV_INIT)) {                       // first trial of a new search: the host begins it with the slope at its start (lbfgs.hpp:756; d = -g after INIT)
                    const double dgi = (flags & DV_ADVANCE) ? ctlD[4] : -gg0;
                    // An ACCEPTED first trial - four rounds in five - is recognised without running the search object (LineSearch::
                    // first_trial_accepted: mt_begin's and mt_feed's own tests on their own operands); it is not needed afterwards, the
                    // next search begins from scratch.  The state machine runs only when this test says no.
                    const bool plain = step != a.ls_min_step && step != a.ls_max_step;
                    accepted_first = (a.fast_control & 1) && plain && LineSearch::first_trial_accepted(lpm, step, f_acc, dgi, fv, dgv);
                    ls_ok = !accepted_first && plain && ls.mt_begin(lpm, step, f_acc, dgi) == 0;
                }
                if (__builtin_expect(accepted_first, 1)) {
                    const int nslot = last_slot < 0 ? 0 : (last_slot + 1 == v.m ? 0 : last_slot + 1), nbound = min(v.m, last_bound + 1);
                    pred_word = ((rk_u64)(nbound & 0xFFF) << 20) | ((rk_u64)(nslot & 0xFFF) << 8) | (rk_u64)(DV_EVAL | DV_ADVANCE | DV_TRIAL | DV_STEP_IS_ONE);
                    pred_kind = 1; spec_ready = true;
                } else if (ls_ok) {
                    const int rc = ls.mt_feed(lpm, fv, dgv);
                    if (rc == LineSearch::PENDING) {
                        if (a.speculate >= 2) {                             // (level 1 follows the search without starting on its next step: see optimize_resident)
                        pred_step = ls.step();
                        pred_word = ((rk_u64)dv_step_hash(pred_step) << 8) | (rk_u64)(DV_EVAL | DV_TRIAL | (pred_step == 1.0 ? DV_STEP_IS_ONE : 0));
                        pred_kind = 2; spec_ready = true;
                        }
                    } else {
                        ls_ok = false;                                      // accepted (the next search begins with the next ADVANCE) or failed (the host falls back to backtracking: its commands are awaited)
                        if (rc > 0) {
                            const int nslot = last_slot < 0 ? 0 : (last_slot + 1 == v.m ? 0 : last_slot + 1), nbound = min(v.m, last_bound + 1);
                            pred_word = ((rk_u64)(nbound & 0xFFF) << 20) | ((rk_u64)(nslot & 0xFFF) << 8) | (rk_u64)(DV_EVAL | DV_ADVANCE | DV_TRIAL | DV_STEP_IS_ONE);
                            pred_kind = 1; spec_ready = true;
                        }
                    }
                }
            }
            RK_PROF(RK_P_WAIT_PART);                                        // (leader: the line search's next command)
            if (spec_ready) seq_pending = hseq;                             // nobody waits for the host's answer to this one: posted behind the next phase word (above)
            else if (t == 0) {
                RoundRes *r = a.h_res + k;
                __hip_atomic_store((rk_u64 *)&r->f, (rk_u64)__double_as_longlong(ctlD[0]), FRX_RLX_SYS);
                __hip_atomic_store((rk_u64 *)&r->dg, (rk_u64)__double_as_longlong(ctlD[1]), FRX_RLX_SYS);
                __hip_atomic_store((rk_u64 *)&r->xx, (rk_u64)__double_as_longlong(ctlD[2]), FRX_RLX_SYS);
                __hip_atomic_store((rk_u64 *)&r->gg, (rk_u64)__double_as_longlong(ctlD[3]), FRX_RLX_SYS);
                __hip_atomic_store((rk_u64 *)&r->dginit, (rk_u64)__double_as_longlong(ctlD[4]), FRX_RLX_SYS);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&r->seq, hseq, FRX_RLX_SYS);
            }
            lstage = 0;
            if (__builtin_expect(spec_ready && pred_kind == 1, 1)) {
                // The predicted ADVANCE goes to the cluster HERE, straight behind the prediction: slot and pair count, the drain of the gradient's copy,
                // the phase word.  The members start on the direction while the leader closes the round, walks back to the top of its loop and takes
                // the step over in its own vectors (timeline, round 4: 1.5 us lay between the prediction and the phase word).
                // (slot and pair count of this step are in `pub` since the previous step's gather - or since INIT for a plan's first step)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                rk_drain_and_meet();
                pseq++;
                if (t == 0) __hip_atomic_store(a.phase + k * RK_WSTRIDE, (pseq << 4) | (unsigned)PH_ADV, FRX_RLX_AGENT);
                adv_published = true;
            } else __syncthreads();
            if (PROF && a.dp.stamps && k == 0 && a.stamp_round > 0 && hseq == (rk_u64)a.stamp_round) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (t < 32) a.dp.stamps[32 + t] = (long long)__hip_atomic_load((unsigned long long *)a.dp.stamps + t, FRX_RLX_AGENT);
            }
            RK_PROF(RK_P_POST);
            break;                                                          // the command is done
        }
        kind = 0;                                                           // the command's own phase is over: pass 1 is its evaluation
        }
        if (leave) break;
    }
    if (PROF && a.prof && t < 16) a.prof[((size_t)k * a.G + v.wg) * 16 + t] = ((rk_u64 *)(sm + L.ctl + 16))[t];
    if (t == 0 && a.spec) { a.spec[c * 4] = n_pred_adv; a.spec[c * 4 + 1] = n_pred_trial; a.spec[c * 4 + 2] = n_redone; }
}

// ============================================================================================================================
// MEMBERS (workgroups 1..G-1): history in registers, penalty share; the last one is also the dense workgroup
// ============================================================================================================================
template <int E, bool PROF, bool RESUME>
__device__ __forceinline__ void rk_member_loop(const RoundArgs &a, RoundView v, const RoundLds &L, double *sm) {
    constexpr int CHT = 2 * E;
    const int k = v.k, wg = v.wg, t = v.t, lane = v.lane, wave = v.wave, m = v.m;
    const bool wt = v.wt;
    const int hg = wg - 1;                                                  // history chunk of this workgroup (workgroups 1 .. G-2)
    rk_ldsword ctlU = (rk_ldsword)(unsigned *)(sm + L.ctl);
    double *ctlD = sm + L.ctl + 8, *sC = sm + L.sC, *yC = sm + L.yC, *gC = sm + L.gC, *xpC = sm + L.xpC, *gpC = sm + L.gpC, *pair = sm + L.pair, *ev = sm + L.role + a.ct_doubles;
    double *pub = v.pub, *part = v.part, *upub = v.upub, *dpub = v.dpub;
    const int slot = t & 127, half = t >> 7;
    rk_u64 prof_last = PROF ? wall_clock64() : 0;
    unsigned tr_n = 0; (void)tr_n;
    double Sreg[E], Yreg[E];
#pragma unroll
    for (int e = 0; e < E; e++) { Sreg[e] = 0.0; Yreg[e] = 0.0; }
    unsigned pseq = 0, nadv = 0, nct = 0;                                   // phases seen; direction phases and evaluation phases among them (the tags of this cluster's granules)
    int jnew = 0, bound = 0;
    bool wipe = false;                                                      // work queue: the next step is the first of a new candidate (workgroup-uniform)
    bool resumed = false; (void)resumed;                                    // RESUME: the take-over INIT has loaded the history
    const double *zC = sm + L.zC;
    if (t < E) sm[L.zC + t] = 0.0;                                          // (ordered before its first use by the barriers of the first phases)
    for (;;) {
        if (t == 0) {
            const rk_u64 dl = wall_clock64() + a.timeout_ticks;
            unsigned w = 0;
            bool ok = true;
            for (unsigned spins = 0;; spins++) {
                w = __hip_atomic_load(a.phase + k * RK_WSTRIDE, FRX_RLX_AGENT);
                if ((w >> 4) == pseq + 1) break;
                // (w >> 4 beyond pseq + 1: this workgroup was held up - preempted, a debugger - for longer than a phase the leader does not wait for, and the
                // word it polls for has been overwritten: fail at once instead of sitting out the round timeout)
                if ((spins & 31u) == 31u && (rk_expired(a, dl) || (int)((w >> 4) - (pseq + 1)) > 0)) { ok = false; break; }
                RK_PAUSE(a);
            }
            if (!ok) { rk_fail(a, RK_ERR_PHASE); w = PH_QUIT; }
            ctlU[0] = w & 15u;
        }
        __syncthreads();
        const int kind = (int)ctlU[0];
        pseq++;
        __syncthreads();
        RK_PROF(RK_P_WAIT_PHASE);
        if (kind == PH_QUIT) break;
        if (kind == PH_NEXT) {                                              // work queue: the cluster takes another candidate - the penalty share follows it
            if (t == 0) ctlU[1] = (unsigned)ldg<true>(pub + 2 * a.NXP);
            __syncthreads();
            const int cn = __builtin_amdgcn_readfirstlane((int)ctlU[1]);
            v.c = cn; v.p0 = a.dp.poff[cn]; v.N = a.dp.poff[cn + 1] - v.p0;
            // The finished plan's pairs must go: passes A and B multiply a slot without a pair by zero instead of selecting (every lane runs them), and a
            // plan that ended on non-finite values (the reference's backtracking search "accepts" NaN objectives; the host gives up after 64 of them)
            // leaves NaN pairs behind - 0 * NaN would poison the new candidate's direction.  They are overwritten with zeros by the new plan's
            // FIRST step, in the same conditional block that stores a step's new pair (a second definition of the 2 E history registers - zeroing them
            // here - sent the E = 56 instantiation to scratch memory).
            wipe = true;
        }
        if (kind == PH_INIT) {                                              // this workgroup's chunk of the start point and its gradient: the first pair's "previous point"
            const int e0 = hg * CHT;
            for (int i = t; i < CHT; i += 256) { xpC[i] = ldg<true>(pub + e0 + i); gpC[i] = ldg<true>(pub + a.NXP + e0 + i); }
            if (RESUME && !resumed) {
                // take-over (RoundArgs::rs): this thread's E elements of s_slot and y_slot from the per-stage history rows (natural element order, zero beyond n);
                // slots without a pair stay zero.  (A second definition of the history registers: this instantiation may spill where the production one does not.)
                resumed = true;
                const int newest = a.rs.newest[k], bnd = a.rs.bound[k];
                int age0 = newest - slot; if (age0 < 0) age0 += m;
                const bool has = slot < m && age0 < bnd;
                const size_t row = ((size_t)v.c * m + (size_t)(slot < m ? slot : 0)) * a.rs.hs;
                const int i0 = hg * CHT + half * E;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const bool in = has && (size_t)(i0 + e) < a.rs.hs;
                    const size_t o = row + (size_t)(in ? i0 + e : 0);
                    const double sv = a.rs.S[o], yv = a.rs.Y[o];
                    Sreg[e] = in ? sv : 0.0; Yreg[e] = in ? yv : 0.0;
                }
            }
        }

        // ------------------------------------------------------------------------------------------------------------------
        // PHASE ADV: new pair into the history, 4 m dot products, dense step, linear combination
        // ------------------------------------------------------------------------------------------------------------------
        if (kind == PH_ADV) {
            nadv++;
            // -- 1. this workgroup's chunk of the accepted point and its gradient; s = x - xp, y = g - gp against the chunk kept from the previous step --
            const int e0 = hg * CHT;
            {   // (thread 0's two control loads used to follow its chunk loads one by one, each behind a wait: three L2 round trips in front of the barrier)
                static_assert(CHT <= 256, "one chunk element per thread");
                const int ic = t < CHT ? t : CHT - 1;
                const double xv = ldg<true>(pub + e0 + ic), gv = ldg<true>(pub + a.NXP + e0 + ic);
                double c1 = 0.0, c2 = 0.0;
                if (t == 0) { c1 = ldg<true>(pub + 2 * a.NXP); c2 = ldg<true>(pub + 2 * a.NXP + 1); }
                if (t < CHT) { sC[t] = xv - xpC[t]; yC[t] = gv - gpC[t]; gC[t] = gv; xpC[t] = xv; gpC[t] = gv; }
                if (t == 0) { ctlU[1] = (unsigned)c1; ctlU[2] = (unsigned)c2; }
            }
            __syncthreads();
            RK_TR(32);                                                          // chunk and control words arrived
            jnew = __builtin_amdgcn_readfirstlane((int)ctlU[1]); bound = __builtin_amdgcn_readfirstlane((int)ctlU[2]);   // wave-uniform by construction
            // -- 2. the new pair replaces slot jnew --
            if (slot == jnew || wipe) {                                     // (wipe: first step of a candidate this cluster took over - every other slot takes zeros, see PH_NEXT)
                const bool mine = slot == jnew;
                const double *ssrc = mine ? sC + half * E : zC, *ysrc = mine ? yC + half * E : zC;
#pragma unroll
                for (int e = 0; e < E; e++) { Sreg[e] = ssrc[e]; Yreg[e] = ysrc[e]; }
            }
            wipe = false;
            // -- 3. pass A: s_j.g, y_j.g, s_j.y_new, y_j.y_new over this thread's elements --
            int age = jnew - slot; if (age < 0) age += m;
            const bool valid = slot < m && age < bound;
            {
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                {
                    // The pair comes from registers, the chunk values (the same for every thread of a half) from the lanes of the thread's own 16-lane row:
                    // register q of gq / yq holds elements 16 q + (lane & 15) of the half's chunk and the FMA takes its operand as a row broadcast
                    // (rk_fma_row).  Rounds 3-4 read each value from LDS as a broadcast read - 2 E reads per thread next to 4 E products, pipelined in
                    // blocks of eight since round 3; now 2 ceil(E / 16) reads.  Same products, same order of additions.  EVERY lane runs the loop (a DPP
                    // operand needs its source lane active): a slot without a pair holds zeros or a finished plan's finite pair, its sums are dropped below.
                    constexpr int NQ = (E + 15) / 16;
                    double gq[NQ], yq[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; q++) { const int el = half * E + min(16 * q + (t & 15), E - 1); gq[q] = gC[el]; yq[q] = yC[el]; }
                    rk_dpp_settle(gq, yq);
                    rk_pass_a_dots(acc, gq, yq, Sreg, Yreg, std::make_integer_sequence<int, E>());
                    if (!valid) { acc[0] = 0.0; acc[1] = 0.0; acc[2] = 0.0; acc[3] = 0.0; }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) pair[(half * 4 + q) * 128 + slot] = acc[q];
            }
            __syncthreads();
            RK_TR(33);                                                          // dot products done
            for (int o = t; o < 512; o += 256) stg<true>(part + (size_t)hg * 512 + o, pair[o] + pair[512 + o], wt);
            rk_drain_and_meet();
            if (t == 0) __hip_atomic_fetch_add(a.cntA + k * RK_WSTRIDE, 1u, FRX_RLX_AGENT);
            RK_PROF(RK_P_PASS_A);
            // -- 5. linear combination d = -gamma g - S u + gamma Y w over this workgroup's elements --
            double coefS, coefY;
            if (t == 0) { const bool ok = rk_wait_eq(a.uflag + k * RK_WSTRIDE, nadv, a); if (!ok) rk_fail(a, RK_ERR_UFLAG); }
            __syncthreads();
            coefS = ldg<true>(upub + slot); coefY = ldg<true>(upub + 128 + slot);
            if (t == 0) ctlD[6] = ldg<true>(upub + 256);
            RK_PROF(RK_P_WAIT_U);
            {
                if (!valid) { coefS = 0.0; coefY = 0.0; }                   // a slot without a pair holds zeros or a finished plan's (finite) pair: 0 s + 0 y = 0, no select per element
                // Element i of the chunk is a sum over the 128 SLOTS of this thread's products - a reduction ACROSS threads for each of the
                // 2 E elements.  Round 2 did it with 14 packed four-value wave reductions per wave (~460 dependent DPP / crossbar
                // instructions, ~1.6 us on a lone wave); now every thread drops its E products into an LDS square [256][E + 1] (the penalty
                // scratch is idle in this phase; odd row stride: conflict-free both ways) and 4 E threads add up one column half each, in
                // slot order - E stores, 64 loads and 64 additions per thread, fixed order.
                double *xt = sm + L.role;                                  // [256][E + 1]
                constexpr int XS = E + 1;
                {
                    double *mine = xt + t * XS;
#pragma unroll
                    for (int e = 0; e < E; e++) mine[e] = coefS * Sreg[e] + coefY * Yreg[e];
                }
                __syncthreads();
                RK_TR(34);                                                      // products in the LDS square
                double *wsum = pair;                                       // [2 halves][2 parts][E]
                if (t < 4 * E) {
                    const int grp = t / E, e = t - grp * E;                // grp = half * 2 + part: slots [64 part, 64 part + 64) of that half
                    const double *col = xt + (size_t)(grp * 64) * XS + e;
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                    for (int i0 = 0; i0 < 64; i0 += 16) {                  // sixteen reads in flight per block (four at a time measured one LDS latency per four additions); same four sums
                        double cb[16];
#pragma unroll
                        for (int j = 0; j < 16; j++) cb[j] = col[(i0 + j) * XS];
                        RK_CHUNK();
#pragma unroll
                        for (int j = 0; j < 16; j += 4) { s0 += cb[j]; s1 += cb[j + 1]; s2 += cb[j + 2]; s3 += cb[j + 3]; }
                        RK_CHUNK();
                    }
                    wsum[grp * E + e] = (s0 + s1) + (s2 + s3);
                }
                __syncthreads();
                RK_TR(35);                                                      // column sums done
                const double gamma = ctlD[6];
                for (int i = t; i < CHT; i += 256) {
                    const int hh = i / E, e = i - hh * E;
                    const double di = (wsum[(2 * hh) * E + e] + wsum[(2 * hh + 1) * E + e]) - gamma * gC[i];
                    rk_ll_put((rk_u64 *)dpub + 2 * (hg * CHT + i), di, nadv, wt);   // granule: the leader polls the elements it gathers
                }
            }
            RK_PROF(RK_P_PASS_B);
        }
        if (kind == PH_CT) { nct++; rk_penalty_share<PROF>(a, v, ev, wg, nct); RK_PROF(RK_P_PENALTY); }
        // ---- report the end of this workgroup's part of the phase to the leader ----
        rk_drain_and_meet();
        if (t == 0) __hip_atomic_fetch_add(a.cntL + k * RK_WSTRIDE, 1u, FRX_RLX_AGENT);
        RK_TR(36);
    }
    if (PROF && a.prof && t < 16) a.prof[((size_t)k * a.G + wg) * 16 + t] = ((rk_u64 *)(sm + L.ctl + 16))[t];
}


// NB: history workgroups whose loads go out as one batch (a cluster has at most 16 workgroups; beyond NB: one load at a time)
template <int NB> __device__ __forceinline__ void rk_gather_partials(const double *part, int nh, int t, double *va, double *vb, double *vc, double *ve) {
    double pv[2][NB];
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
        for (int w2 = 0; w2 < NB; w2++) pv[h2][w2] = ldg<true>(part + (size_t)(w2 < nh ? w2 : nh - 1) * 512 + t + 256 * h2);
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
        const int o = t + 256 * h2;
        double s = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NB; w2++) s += w2 < nh ? pv[h2][w2] : 0.0;               // (the select outside the chain of additions)
        for (int w2 = NB; w2 < nh; w2++) s += ldg<true>(part + (size_t)w2 * 512 + o);
        (o < 128 ? va : o < 256 ? vb : o < 384 ? vc : ve)[o & 127] = s;
    }
}

// ============================================================================================================================
// DENSE (workgroup G-1): the m x m part of the compact representation, nothing else
//   R^-1 in LDS as a full [128][129] square indexed by SLOT (entry (i, j) is non-zero only when pair i is not newer than pair j;
//   the odd row stride makes a row sweep and a column sweep both conflict-free, and the zeros replace every age test), Y^T Y in
//   registers (thread (p, hq) holds row p, columns [64 hq, 64 hq + 64)).  When the oldest pair is dropped R^-1 loses that row and
//   column and nothing else changes; a new pair appends the column (-R22^-1 c / rho, 1 / rho), c = S^T y_new, rho = s_new . y_new.
//   Checked on a complete headline optimisation (3500 accepted steps, cond(R) up to 1e6): the direction stays within 3e-13 of the
//   two-loop recursion, no drift (scripts/lbfgs_inverse_stability.py).
// ============================================================================================================================
template <bool PROF, bool RESUME>
__device__ __forceinline__ void rk_dense_loop(const RoundArgs &a, const RoundView &v, const RoundLds &L, double *sm) {
    const int k = v.k, t = v.t;
    const bool wt = v.wt;
    const int nh = a.G - 2;                                                 // history workgroups
    rk_ldsword ctlU = (rk_ldsword)(unsigned *)(sm + L.ctl);
    double *Rf = sm + L.Rf, *vd = sm + L.vd, *va = sm + L.va, *vb = sm + L.vb, *vc = sm + L.vc, *ve = sm + L.ve, *vw = sm + L.vw, *vv = sm + L.vv,
           *mv = sm + L.mv, *mz = sm + L.mv + 256;
    double *pub = v.pub, *part = v.part, *upub = v.upub;
    double *ctlD = sm + L.ctl + 8;
    const int pp = t & 127, hq = t >> 7, q0 = 64 * hq;
    rk_u64 prof_last = PROF ? wall_clock64() : 0;
    unsigned tr_n = 0; (void)tr_n;
    double Ya[32], Yb[32];                                                  // (Y^T Y)[pp][q0 .. q0 + 31], [q0 + 32 .. q0 + 63]: two arrays the compiler keeps in registers
#pragma unroll
    for (int u = 0; u < 32; u++) { Ya[u] = 0.0; Yb[u] = 0.0; }
    for (int i = L.Rf + t; i < L.mv + 512; i += 256) sm[i] = 0.0;           // no uninitialised word is ever multiplied
    __syncthreads();
    unsigned pseq = 0, nadv = 0;
    bool resumed = false; (void)resumed;
    // (An evaluation phase is acknowledged by thread 0 alone, inside its poll loop, while the other waves stay parked at the barrier: nothing of it concerns
    // this workgroup.  Round 4 also tried a FORWARDER here - a lane of the idle workgroup copying the host's command from the mapped mailbox into
    // device memory for the leader, to take the PCIe read and its tail out of the adjoint: the adjoint became flat (5.4 us) but the confirmation
    // waited more often, 0.8 us per round slower in all; what the tail really was - mailbox threads on the other socket - is cured on the host
    // side, device_numa_cpus in frx_api.cpp: profiles/r04_numa.txt.)
    for (;;) {
        if (t == 0) {
            const rk_u64 dl = wall_clock64() + a.timeout_ticks;
            unsigned w = 0;
            bool ok = true;
            for (unsigned spins = 0;; spins++) {
                w = __hip_atomic_load(a.phase + k * RK_WSTRIDE, FRX_RLX_AGENT);
                if ((w >> 4) == pseq + 1) {
                    if ((w & 15u) != (unsigned)PH_CT) break;
                    pseq++;                                                 // an evaluation phase: nothing to do here but to report (no payload, no barrier)
                    __hip_atomic_fetch_add(a.cntL + k * RK_WSTRIDE, 1u, FRX_RLX_AGENT);
                    spins = 0;
                    continue;
                }
                if ((spins & 31u) == 31u && (rk_expired(a, dl) || (int)((w >> 4) - (pseq + 1)) > 0)) { ok = false; break; }   // (expired, or overrun: see the members' loop)
                RK_PAUSE(a);
            }
            if (!ok) { rk_fail(a, RK_ERR_PHASE); w = PH_QUIT; }
            ctlU[0] = w & 15u; ctlU[6] = pseq + 1;
        }
        __syncthreads();
        const int kind = (int)ctlU[0];
        pseq = ctlU[6];
        __syncthreads();
        RK_PROF(RK_P_WAIT_PHASE);
        if (kind == PH_QUIT) break;
        if (kind == PH_NEXT) {                                              // work queue: a new plan starts with an empty history
#pragma unroll
            for (int u = 0; u < 32; u++) { Ya[u] = 0.0; Yb[u] = 0.0; }
            for (int i = L.Rf + t; i < L.mv + 512; i += 256) sm[i] = 0.0;
            __syncthreads();
        }
        if (RESUME && kind == PH_INIT && !resumed) {                         // take-over (RoundArgs::rs): R^-1 by slot, D and Y^T Y as the host rebuilt them from the per-stage history
            resumed = true;
            const double *ri = a.rs.rinv + (size_t)k * 128 * RK_RS;
            for (int i = t; i < 128 * RK_RS; i += 256) Rf[i] = ri[i];
            if (t < 128) vd[t] = a.rs.vd[(size_t)k * 128 + t];
            const double *yy = a.rs.yy + ((size_t)k * 128 + pp) * 128 + q0;
#pragma unroll
            for (int u = 0; u < 32; u++) { Ya[u] = yy[u]; Yb[u] = yy[32 + u]; }
            __syncthreads();
        }
        if (kind == PH_ADV) {
            nadv++;
            // Row pp of R^-1 as the previous step left it goes into registers NOW, while thread 0 still waits for the partial sums (round 4: pass 1 used
            // to read it from LDS in chunks of 16 in front of their products, one exposed LDS latency plus the four waves' shared read bandwidth per
            // chunk - 1.44 us for 0.6 us of products).  The entry of column jnew is stale then (the dropped pair's): the broadcast vectors carry a zero there.
            double er[64];
            {
                const double *row = Rf + pp * RK_RS + q0;
#pragma unroll
                for (int j = 0; j < 64; j++) er[j] = row[j];
            }
            RK_CHUNK();
            if (t == 0) { ctlU[1] = (unsigned)ldg<true>(pub + 2 * a.NXP); const bool ok = rk_wait_eq(a.cntA + k * RK_WSTRIDE, (unsigned)nh * nadv, a); if (!ok) rk_fail(a, RK_ERR_DENSE); }
            __syncthreads();
            const int jnew = __builtin_amdgcn_readfirstlane((int)ctlU[1]);
            RK_PROF(RK_P_WAIT_PART);
            // partial sums of the history workgroups, summed in workgroup order (deterministic); all loads of a batch in flight together.  Six history
            // workgroups (the full chip at the headline size) take the short form: 12 loads and additions per thread instead of 28 - the batch of 14 padded
            // with repeats of the last workgroup's address measured 1.18 us from the count to the sums in LDS, the exact batch 0.60 us.
            if (nh <= 6) rk_gather_partials<6>(part, nh, t, va, vb, vc, ve);
            else if (nh <= 12) rk_gather_partials<12>(part, nh, t, va, vb, vc, ve);
            else rk_gather_partials<14>(part, nh, t, va, vb, vc, ve);
            if (t < 128) { Rf[jnew * RK_RS + t] = 0.0; Rf[t * RK_RS + jnew] = 0.0; }                  // the pair that slot jnew held is gone
            __syncthreads();
            RK_PROF(RK_P_DENSE_IN);
            const int ln = t & 63, l16 = t & 15;                            // every 16-lane row holds elements q0 + 16 q + l16 of a vector in register q (rk_fma_row)
            // (Y^T Y's registers take the new pair's row and column AFTER the result is published - 0.8 us off the critical path; until
            // then row / column jnew of the registers still hold the dropped pair: pass 2 masks them and adds the new ones from `ve`)
            const double rho = vc[jnew], gamma = rho / ve[jnew], wl = va[jnew] / rho;      // y.s, y.s / y.y of the newest pair (lbfgs.hpp:1403)
            // pass 1: rows of the old R^-1, two right-hand sides at once: z = R22^-1 c, tt = R22^-1 a (row and column jnew are zero)
            {
                double sz = 0.0, st = 0.0, xc[4], xa[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int el = q0 + 16 * q + l16;
                    const bool dropped = el == jnew;                        // column jnew of the old R^-1 counts as zero
                    xc[q] = dropped ? 0.0 : vc[el]; xa[q] = dropped ? 0.0 : va[el];
                }
                rk_dpp_settle(xc, xa);
#pragma unroll
                for (int q = 0; q < 4; q++) rk_dot16x2(sz, st, xc[q], xa[q], er + 16 * q, rk_seq16());
                mv[hq * 128 + pp] = sz; mz[hq * 128 + pp] = st;
            }
            __syncthreads();
            RK_PROF(RK_P_PASS_A);                                           // (dense workgroup: pass 1)
            if (t < 128) {                                                  // new column of R^-1, and w = R^-1 a
                double wp;
                if (pp == jnew) { Rf[pp * RK_RS + jnew] = 1.0 / rho; wp = wl; vd[jnew] = rho; }
                else { const double z = mv[pp] + mv[128 + pp]; Rf[pp * RK_RS + jnew] = -z / rho; wp = (mz[pp] + mz[128 + pp]) - z * wl; }
                vw[t] = wp;
            }
            __syncthreads();
            // pass 2: (Y^T Y) w.  The registers hold the matrix of the PREVIOUS step; the new pair's row and column are e = Y^T y_new (ve):
            //   (YY w)[p] = sum_{q != jnew} YY_old[p][q] w[q] + e[p] w[jnew]   (p != jnew),      (YY w)[jnew] = e . w
            double ec[64];                                                  // column pp of the NEW R^-1, rows q0 .. q0 + 63: on its way during pass 2, used by pass 3
            {
                double sacc = 0.0, xw[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { const int el = q0 + 16 * q + l16; xw[q] = el == jnew ? 0.0 : vw[el]; }
                const double vet = t < 128 ? ve[t] : 0.0, vwt = t < 128 ? vw[t] : 0.0;
                RK_CHUNK();
                {
                    const double *col = Rf + q0 * RK_RS + pp;
#pragma unroll
                    for (int j = 0; j < 64; j++) ec[j] = col[j * RK_RS];
                }
                RK_CHUNK();
                if (t < 128) {                                              // e . w on waves 0 and 1, interleaved with the products below
                    const double ws = wave_sum_dpp(vet * vwt);
                    if (ln == 0) ctlD[t >> 6] = ws;
                }
                rk_dpp_settle(xw);
                rk_dot16(sacc, xw[0], Ya, rk_seq16()); rk_dot16(sacc, xw[1], Ya + 16, rk_seq16());
                rk_dot16(sacc, xw[2], Yb, rk_seq16()); rk_dot16(sacc, xw[3], Yb + 16, rk_seq16());
                mv[hq * 128 + pp] = sacc;
            }
            __syncthreads();
            RK_PROF(RK_P_FORWARD);                                          // (dense workgroup: column update + pass 2)
            if (t < 128) {
                const double yyw = t == jnew ? ctlD[0] + ctlD[1] : (mv[t] + mv[128 + t]) + ve[t] * vw[jnew];
                vv[t] = vd[t] * vw[t] + gamma * yyw - gamma * vb[t];
            }
            __syncthreads();
            // pass 3: columns of the new R^-1: u = R^-T v
            {
                double sacc = 0.0, xv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) xv[q] = vv[q0 + 16 * q + l16];
                rk_dpp_settle(xv);
#pragma unroll
                for (int q = 0; q < 4; q++) rk_dot16(sacc, xv[q], ec + 16 * q, rk_seq16());
                mz[hq * 128 + pp] = sacc;
            }
            __syncthreads();
            RK_PROF(RK_P_PASS_B);                                           // (dense workgroup: pass 3)
            // (Round 4 tried this hand-off as granules polled by every history thread: no faster at 32 candidates, 0.25 us per round slower with twelve history
            // workgroups per cluster - 3072 polling threads instead of 12; the flag stayed.  profiles/NOTES.md)
            if (t < 128) {
                stg<true>(upub + t, -(mz[t] + mz[128 + t]), wt);
                stg<true>(upub + 128 + t, gamma * vw[t], wt);
            }
            if (t == 128) stg<true>(upub + 256, gamma, wt);
            rk_drain_and_meet();
            if (t == 0) __hip_atomic_store(a.uflag + k * RK_WSTRIDE, nadv, FRX_RLX_AGENT);
            RK_PROF(RK_P_SOLVE);
            {   // Y^T Y: row and column jnew (zero where there is no pair: the partial sums are) - behind the publication
                const int uj = jnew - q0;
                const double colv = ve[pp];
                const bool isrow = pp == jnew;
#pragma unroll
                for (int u = 0; u < 32; u++) {                              // (LDS broadcasts here: with rk_bcast this loop measured 1.3 instead of 0.8 us)
                    const double r0 = ve[q0 + u], r1 = ve[q0 + 32 + u];
                    Ya[u] = isrow ? r0 : (u == uj ? colv : Ya[u]);
                    Yb[u] = isrow ? r1 : (u + 32 == uj ? colv : Yb[u]);
                }
            }
            RK_PROF(RK_P_VECTORS);                                          // (dense workgroup: Y^T Y update)
        }
        rk_drain_and_meet();
        if (t == 0) __hip_atomic_fetch_add(a.cntL + k * RK_WSTRIDE, 1u, FRX_RLX_AGENT);
    }
    if (PROF && a.prof && t < 16) a.prof[((size_t)k * a.G + v.wg) * 16 + t] = ((rk_u64 *)(sm + L.ctl + 16))[t];
}

// NR: 64 = every candidate has <= 64 pieces (the wave-specialised bodies, nothing else compiled in), 0 = the geometry class is a run-time value
// ARGP (round 6): the launch's arguments - ~90 scalar registers' worth of pointers and sizes for a budget of 102 - come through a pointer to a copy
// in device memory instead of by value: a field is then an s_load at its use (scalar cache) instead of a register that is live for the whole plan and spilled to a
// VGPR lane (production instantiation: 602 -> 301 SGPR spills, 4077 -> 1377 v_readlane reloads, 19.0 k -> 16.2 k VALU instructions; scripts/isa_report.py).
template <int E, bool PROF, int NR, bool RESUME = false, bool ARGP = false>
__global__ __launch_bounds__(256, 1) void k_round(typename std::conditional<ARGP, const RoundArgs *__restrict__, RoundArgs>::type arg) {
    const RoundArgs &a = [&]() -> const RoundArgs & { if constexpr (ARGP) return *arg; else return arg; }();
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int CHT = 2 * E;
    // Workgroup -> (candidate, role).  Blocks b with equal b % 8 have been observed to share an XCD (MI355X guide; HIP promises
    // nothing), so cluster c takes the blocks (c % 8) + 8 (wg + G (c / 8)): if the observation holds, a cluster's hand-offs stay
    // inside one XCD's L2.  Nothing relies on it: every workgroup publishes its XCC_ID, and only a cluster that finds all of
    // its members on one XCD switches its payload stores from write-through to plain (`wt`).
    const int lane8 = blockIdx.x & 7, rest = blockIdx.x >> 3;
    RoundView v;
    v.wg = rest % a.G; v.k = lane8 + 8 * (rest / a.G);
    v.t = threadIdx.x; v.lane = v.t & 63; v.wave = v.t >> 6;
    if (v.k >= a.S) return;                                               // grid is 8 G ceil(S / 8) blocks
    v.c = RESUME ? a.rs.cand[v.k] : v.k;                                  // cluster k starts on candidate k (RESUME: on the straggler it takes over); with more candidates than clusters the host hands out the rest (DV_NEXT)
    v.m = a.m;
    const RoundLds L = round_lds(a.m, CHT, a.eval_doubles);
    rk_ldsword ctlU = (rk_ldsword)(unsigned *)(sm + L.ctl);          // [0] ok / kind, [1..2] command word / slot, pair count, [3] one XCD
    v.xbase = a.dp.xoff[v.c]; v.n = a.dp.xoff[v.c + 1] - v.xbase;
    v.p0 = a.dp.poff[v.c]; v.N = a.dp.poff[v.c + 1] - v.p0;
    v.pub = a.pubsyg + (size_t)v.k * (3 * a.NXP + 2); v.part = a.part + (size_t)v.k * a.G * 512; v.upub = a.upub + (size_t)v.k * 258; v.dpub = a.dpub + (size_t)v.k * 2 * a.NXP;
    // ---- census: every workgroup of the launch must be resident before anybody waits for anybody ----
    if (v.t == 0) {
        unsigned my_xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
        my_xcc = (my_xcc & 15u) + 1u;
        unsigned hw_id = 0;                                                 // where this workgroup runs (diagnostic: bits 8.. of the word; CU_ID 11:8, SH_ID 12, SE_ID 15:13 of HW_ID)
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        __hip_atomic_store(a.xcc + v.k * a.G + v.wg, my_xcc | ((hw_id & 0xFF00u) << 0), FRX_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(a.census, 1u, FRX_RLX_AGENT);
        bool ok;
        {   // the census has its own, shorter bound: a chip that cannot host the whole grid at once is a configuration, not a fault
            const rk_u64 dl = wall_clock64() + a.census_ticks;
            for (unsigned spins = 0;; spins++) {
                if (__hip_atomic_load(a.census, FRX_RLX_AGENT) == (unsigned)(a.S * a.G)) { ok = true; break; }
                if ((spins & 31u) == 31u && rk_expired(a, dl)) { ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (!ok) {
            // Tell the HOST, which can poll the mailboxes but not the status word: without this its service loop sat out the whole round
            // timeout (5 s against the 250 ms census bound) and frx_optimize failed hard instead of taking the per-stage path.
            rk_fail(a, RK_ERR_CENSUS);
            if (v.wg == 0) __hip_atomic_store(&a.h_res[v.k].seq, ~(rk_u64)0, FRX_RLX_SYS);
        }
        bool same = ok && !(a.fast_control & 8);                            // (FRX_RESIDENT_WRITE_THROUGH=1: every cluster takes the cross-XCD form of its hand-offs - tests)
        for (int k = 0; k < a.G; k++) same = same && (__hip_atomic_load(a.xcc + v.k * a.G + k, FRX_RLX_AGENT) & 0xFFu) == my_xcc;
        ctlU[0] = ok ? 1u : 0u;
        ctlU[3] = same ? 1u : 0u;
    }
    __syncthreads();
    if (ctlU[0] == 0u) return;
    v.wt = ctlU[3] == 0u;                                                 // write-through payload stores unless the cluster shares an XCD
    if (PROF && v.t < 16) ((rk_u64 *)(sm + L.ctl + 16))[v.t] = 0;
    __syncthreads();
    if (v.wg == 0) rk_leader_loop<PROF, NR, RESUME>(a, v, L, sm);
    else if (v.wg == a.G - 1) rk_dense_loop<PROF, RESUME>(a, v, L, sm);
    else rk_member_loop<E, PROF, RESUME>(a, v, L, sm);
}
#undef RK_PROF
#undef RK_TR

} // namespace frx
