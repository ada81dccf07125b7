// HIP kernels (gfx950 / CDNA4, wave64) for one batched objective evaluation
//     x  →  (f, ∇f)          = SE3GCOPTER::objectiveFunc, CPU.hpp:961-1000, for B candidates.
//
//   k_forward   one 64-lane workgroup per candidate: tau→T, xi→q, band assembly, no-pivot banded
//               LU + 3-column solve with the band resident in LDS        (CPU.hpp:626-676, 729-747, 425-505; traj.hpp:655-719)
//   k_penalty   the hot kernel: every (candidate, piece, sample) of the batch; lanes = samples,
//               corridor polytopes + coefficients staged in LDS, fixed-order LDS reduction
//               of the 20 partials per piece                               (CPU.hpp:188-408)
//   k_backward  one workgroup per candidate: jerk cost/gradients, adjoint banded solve,
//               propagation to T and q, diffeomorphism layers               (CPU.hpp:507-520, 65-161, 816-928; traj.hpp:724-751)
//
// HBM layout (all FP64):
//   C      [P][6][3]    piece-major coefficients, 144 B per piece (P = fine pieces of the whole batch)
//   T      [P]          piece durations
//   hblk   [P][Kmax+1][4]  per-piece corridor block: {origin xyz, K} then K x (unit normal, c = n.(p - org) - margin)
//   out20  [P][20]      per-piece partials {cost, gdT, gdC[6][3]} written by k_penalty
//   band   [sum 6N_b*13] LU factors per candidate, row-window layout A(i,j) → [i*13 + (j-i+6)]
//   x, g   packed per candidate (tau[dimT], xi[...]);  f [B]
#pragma once
#include <hip/hip_runtime.h>

#include "frx_device.hpp"
#include "frx_minco.hpp"
#include "frx_lbfgs.hpp"
#include "frx_wave.hpp"

// Linkage of the stage kernels defined in this header: external in the translation unit that launches them (frx_device.hip); the translation units that only
// want the BODIES (frx_device_round.hip, frx_device_eval.hip) define it as `static`, and the unused kernels are not compiled there at all.
#ifndef FRX_KERNEL_LINKAGE
#define FRX_KERNEL_LINKAGE
#endif

namespace frx {

// Optional epilogue of k_backward_knot for the device-vector L-BFGS: with d (the search direction) given, the kernel also
// reduces g.d, x.x and g.g and, for candidates whose command carries DV_EVAL, writes the round's DvResult, which saves the
// separate k_lbfgs_post launch.  `flags` is the device copy of the commands' flags that k_lbfgs_pre leaves behind (reading the
// command itself would put a PCIe round trip in front of the kernel).  d == nullptr: plain objective evaluation.
// `arrive`/`flag`/`round`: completion mailbox.  Every workgroup bumps the device counter after its result is visible system-wide;
// the one that brings it to B * round writes `round` into a word of mapped host memory, on which the host spins instead of
// polling the stream through the driver (the reference's cuda_computer signals completion the same way, cc.cu:384-405, 537-547).
// `lds_out` (persistent round kernel): the four values go to this LDS array {f, g.d, x.x, g.g} instead of `res`, whatever the flags.
// `early_cmd` (resident round kernel, optional): a command {word, step} in mapped HOST memory that thread 0 reads while the adjoint runs and leaves in lds_out[7], lds_out[6] (bit
// patterns; word first: the host writes the step first and the word last, so a word with the expected sequence number comes with its step) - the host's
// command for a round the leader started on a prediction.  The read crosses PCIe (1-2.5 us with 32 leaders polling): it is issued only AFTER thread 0's wave has consumed its
// first batch of loads - vmcnt completes in order, so issued in front of them it held up the wave's whole chain by the PCIe round trip (measured: adjoint 6.5 us with one
// candidate on the chip, 7.5 us with 32).
struct LineSearchTap { const double *d; const int *flags; DvResult *res; unsigned *arrive; volatile unsigned *flag; unsigned round; double *lds_out = nullptr; const unsigned long long *early_cmd = nullptr; };



// Inter-workgroup data inside ONE launch (the persistent round kernel, frx_round_kernel.hpp): a CU's L1 is never refreshed by
// another CU's stores and the per-XCD L2s are not coherent with each other (MI355X guide, "inter-workgroup visibility").  Payload
// that crosses workgroups is therefore written with write-through (sc1) stores and read with L1-bypassing (sc1) loads - relaxed
// agent-scope atomics lower to exactly these - and ordered by a drained flag / counter (Guideline 16, form R1).  SH = false: the
// plain accesses of the one-launch-per-stage kernels (a kernel boundary orders everything).
template <bool SH> __device__ __forceinline__ double ldg(const double *p) {
    if (SH) return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}
// `wt` (run-time, SH only): false when the launch has VERIFIED (XCC_ID census of the cluster, frx_round_kernel.hpp) that producer and
// consumers sit on one XCD - then the shared L2 is the coherence point and a plain store (L1 is write-through) that the consumer reads
// with an L1-bypassing load is enough; measured 30-40 % off every hand-off (profiles/r02_cluster_probe.txt).
template <bool SH> __device__ __forceinline__ void stg(double *p, double v, bool wt = true) {
    if (SH && wt) __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// ---- hand-offs without a flag: self-validating granules (resident round kernel) -----------------------------------------------------------------
// A double that another workgroup waits for can travel as TWO 8-byte words {low half | tag << 32}, {high half | tag << 32} (the LL form of the
// collectives libraries): the consumer polls the payload itself and takes a value when both words carry the tag it expects (the number of the
// phase - never repeated within a launch; the buffers start zeroed).  Each word is one naturally atomic 8-byte access, so nothing is
// drained, no flag follows, no workgroup meets: one L2 round trip where flag-then-payload (drain, barrier, flag; poll, barrier, payload loads)
// takes two and a half.  Used where ONE workgroup is the consumer and each of its threads polls a few granules (history -> leader: 3 per thread, penalty
// partials -> adjoint: 2 or 6).  Where many workgroups would poll (the dense result: every thread of 6-12 history workgroups) or a thread many granules
// (partial sums into the dense workgroup: 2 x 6..14), the polls cost more than the flag - measured - and the drained flag / counter stayed.
typedef unsigned long long ll_u64;
__device__ __forceinline__ void rk_ll_put(ll_u64 *slot, double v, unsigned tag, bool wt) {
    const ll_u64 b = (ll_u64)__double_as_longlong(v), tg = (ll_u64)tag << 32;
    const ll_u64 w0 = (b & 0xFFFFFFFFull) | tg, w1 = (b >> 32) | tg;
    if (wt) { __hip_atomic_store(slot, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(slot + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else { slot[0] = w0; slot[1] = w1; }
}
__device__ __forceinline__ bool rk_ll_ok(ll_u64 w0, ll_u64 w1, unsigned tag) { return (unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag; }
__device__ __forceinline__ double rk_ll_value(ll_u64 w0, ll_u64 w1) { return __longlong_as_double((long long)((w0 & 0xFFFFFFFFull) | (w1 << 32))); }

// Operands that a RESIDENT caller (frx_round_kernel.hpp, leader workgroup) keeps in LDS from round to round, so that the evaluation
// bodies neither stage them from global memory nor send the reduction multipliers through it: xs = the candidate's variables,
// vs = its waypoint polytopes, dsv = the search direction, pw = [nrow][8 steps + 5] multipliers (only used by the wave-specialised
// reduction, nrow == 64), gs = where the gradient goes, wq (optional, [nrow][4]) = per waypoint {|xi|^2, sum_a V_a xi_a^2} left by the forward map for the
// adjoint of the same evaluation (its first pass over the vertices then is three multiplications).  vskew = 1: the polytope of waypoint w starts w doubles further on than in the
// packed layout.  The blocks of consecutive waypoints are 3 nv doubles apart (36 for the 12-vertex overlaps of box corridors: a
// multiple of 4 - the lanes of a wave, one waypoint per lane pair, then hit 8 of the 32 LDS double-banks, a four-way conflict on
// every read of the waypoint map and of its adjoint, measured 2.7 k cycles for a 12-vertex pass); one double of skew makes the stride odd.  nullptr (the one-launch-per-stage kernels): everything is staged per call, as before.
struct ResidentOps { double *xs, *vs, *dsv, *pw, *gs = nullptr; int vskew = 0; double *wq = nullptr; double *gpub = nullptr; bool gwt = true;
                     const ll_u64 *o20ll = nullptr; unsigned o20tag = 0; unsigned *status = nullptr; ll_u64 spin_ticks = 0;
                     bool quiet = false; };      // quiet: no cycle stamps from this caller (the members of the one-launch evaluation run the forward map of candidate 0 too)   // gs (optional): the gradient goes to this LDS array INSTEAD of g; gpub (optional, global): and to this array, for the other workgroups of the cluster (write-through unless gwt is false)

// What the forward map reads from the problem's index tables at its top, per candidate and per thread: offsets, this thread's piece (coarse index, interval
// count), its pair's waypoint (vertex count, first vertex, first variable), the fixed end states of its axis.  Constant for the length of a plan: a RESIDENT
// caller fetches them once (rk_leader_loop, load_candidate) and hands them in; fetched inside the body they cost it a scalar load and a dependent vector
// load from L2 in front of everything else - 1 100 cycles before the first duration is formed (cycle stamps, round 5).
struct KnotPre { int p0, N, c0, cN, x0, cv0, pc, piv, wnv, wvb, wxb; double bs[6]; };
// The one-launch evaluation's way out of the forward map (forward_knot_body<.., MODE & 2>: frx_eval_kernel.hpp): (C, T) leave as granules tagged `tag` in ll ([P][19],
// duration at index 18) INSTEAD of plain stores, and `gate` (optional) receives gate_val (the tag and the leader's XCD) right behind them.  mxw: nmx words in which the
// consumers published gate_val if they run on the leader's XCD - when all did, the granules leave as plain stores (that XCD's L2 is the meeting point) instead of
// write-through ones; null: always write-through.
struct GranuleOut { ll_u64 *ll; unsigned tag; unsigned *gate; unsigned gate_val; const unsigned *mxw; int nmx; };

// Coalesced staging global -> LDS with every load of a trip in flight before the first LDS store.  The plain loop
// `for (i = k; i < n; i += nthr) dst[i] = src[i]` compiles to load / s_waitcnt vmcnt(0) / ds_write per element even under
// `#pragma unroll 8` (one full memory latency per element and thread: 9.4k of the adjoint's 30k cycles, measured); clamped,
// unconditional loads into a register block let the scheduler issue the whole trip at once.
template <int U> __device__ __forceinline__ void stage_to_lds(double *dst, const double *__restrict__ src, int n, int k, int nthr) {
    for (int i0 = k; i0 < n; i0 += U * nthr) {
        double tmp[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * nthr; tmp[u] = src[i < n ? i : n - 1]; }
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * nthr; if (i < n) dst[i] = tmp[u]; }
    }
}

__device__ __forceinline__ double wave_sum(double v) {
    // butterfly: every lane ends with the same, order-fixed sum of the 64 lane values
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define BAND(i, j) band[(i) * FRX_BAND_W + ((j) - (i) + 6)]

// ---------------------------------------------------------------------------------------------
// k_forward: grid = B, block = 64.  Dynamic LDS: band[6N*13] | rhs[6N*3] | Tf[N] | Tc[cN]
// ---------------------------------------------------------------------------------------------
FRX_KERNEL_LINKAGE __global__ __launch_bounds__(64) void k_forward(DevProblem dp, const double *__restrict__ x, double *__restrict__ Tout,
                                                double *__restrict__ Cout, double *__restrict__ bandOut, int maxN, int maxCN) {
    extern __shared__ double sm[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int p0 = dp.poff[b], N = dp.poff[b + 1] - p0, n6 = 6 * N;
    const int c0 = dp.coff[b], cN = dp.coff[b + 1] - c0;
    const int x0 = dp.xoff[b];
    double *band = sm;
    double *rhs = band + (size_t)6 * maxN * FRX_BAND_W;
    double *Tf = rhs + (size_t)6 * maxN * 3;
    double *Tc = Tf + maxN;

    // ---- forwardT (CPU.hpp:626-676) ----
    if (dp.soft) {
        for (int i = lane; i < cN; i += 64) Tc[i] = tau_to_T(x[x0 + i], dp.c2 != 0);
    } else {
        if (lane == 0) {
            const int Ms1 = cN - 1;
            double sum = 0.0;
            for (int i = 0; i < Ms1; i++) { Tc[i] = tau_to_T(x[x0 + i], dp.c2 != 0); }
            Tc[Ms1] = 0.0;
            for (int i = 0; i <= Ms1; i++) sum += Tc[i];
            const double den = 1.0 + sum;
            for (int i = 0; i <= Ms1; i++) Tc[i] /= den;
            sum = 0.0;
            for (int i = 0; i <= Ms1; i++) sum += Tc[i];
            Tc[Ms1] = 1.0 - sum;
            for (int i = 0; i <= Ms1; i++) Tc[i] *= dp.sumT;
        }
    }
    for (int i = lane; i < n6 * FRX_BAND_W; i += 64) band[i] = 0.0;
    for (int i = lane; i < n6 * 3; i += 64) rhs[i] = 0.0;
    __syncthreads();
    // ---- splitToFineT (CPU.hpp:930-944) ----
    for (int i = lane; i < N; i += 64) {
        const int gc = dp.piece_coarse[p0 + i];
        const double t = Tc[gc - c0] / dp.coarse_iv[gc];
        Tf[i] = t;
        Tout[p0 + i] = t;
    }
    __syncthreads();

    // ---- forwardP (CPU.hpp:729-747) straight into rhs row 6i+5, and band assembly (CPU.hpp:437-499) ----
    const int w0 = p0 - b;
    for (int i = lane; i < N - 1; i += 64) {
        const int gw = w0 + i;
        const int k = dp.wp_nv[gw] - 1;
        const double *V = dp.vrec + 3 * (size_t)dp.wp_vbeg[gw];
        const double *xi = x + dp.wp_xbeg[gw];
        double nrm = 0.0;
        for (int a = 0; a < k; a++) nrm += xi[a] * xi[a];
        const double sc = 2.0 / (1.0 + nrm);
        double q0 = 0.0, q1 = 0.0, q2 = 0.0;
        for (int a = 0; a < k; a++) {
            const double r = sc * xi[a], rr = r * r;
            q0 += V[3 * (a + 1)] * rr;
            q1 += V[3 * (a + 1) + 1] * rr;
            q2 += V[3 * (a + 1) + 2] * rr;
        }
        rhs[(6 * i + 5) * 3 + 0] = q0 + V[0];
        rhs[(6 * i + 5) * 3 + 1] = q1 + V[1];
        rhs[(6 * i + 5) * 3 + 2] = q2 + V[2];

        const double t1 = Tf[i], t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
        const int r = 6 * i;
        BAND(r + 3, r + 3) = 6.0;  BAND(r + 3, r + 4) = 24.0 * t1; BAND(r + 3, r + 5) = 60.0 * t2; BAND(r + 3, r + 9) = -6.0;
        BAND(r + 4, r + 4) = 24.0; BAND(r + 4, r + 5) = 120.0 * t1; BAND(r + 4, r + 10) = -24.0;
        BAND(r + 5, r) = 1.0; BAND(r + 5, r + 1) = t1; BAND(r + 5, r + 2) = t2; BAND(r + 5, r + 3) = t3; BAND(r + 5, r + 4) = t4; BAND(r + 5, r + 5) = t5;
        BAND(r + 6, r) = 1.0; BAND(r + 6, r + 1) = t1; BAND(r + 6, r + 2) = t2; BAND(r + 6, r + 3) = t3; BAND(r + 6, r + 4) = t4; BAND(r + 6, r + 5) = t5;
        BAND(r + 6, r + 6) = -1.0;
        BAND(r + 7, r + 1) = 1.0; BAND(r + 7, r + 2) = 2 * t1; BAND(r + 7, r + 3) = 3 * t2; BAND(r + 7, r + 4) = 4 * t3; BAND(r + 7, r + 5) = 5 * t4;
        BAND(r + 7, r + 7) = -1.0;
        BAND(r + 8, r + 2) = 2.0; BAND(r + 8, r + 3) = 6 * t1; BAND(r + 8, r + 4) = 12 * t2; BAND(r + 8, r + 5) = 20 * t3;
        BAND(r + 8, r + 8) = -2.0;
    }
    if (lane == 0) {
        BAND(0, 0) = 1.0; BAND(1, 1) = 1.0; BAND(2, 2) = 2.0;
        const double t1 = Tf[N - 1], t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
        BAND(n6 - 3, n6 - 6) = 1.0; BAND(n6 - 3, n6 - 5) = t1; BAND(n6 - 3, n6 - 4) = t2; BAND(n6 - 3, n6 - 3) = t3; BAND(n6 - 3, n6 - 2) = t4; BAND(n6 - 3, n6 - 1) = t5;
        BAND(n6 - 2, n6 - 5) = 1.0; BAND(n6 - 2, n6 - 4) = 2 * t1; BAND(n6 - 2, n6 - 3) = 3 * t2; BAND(n6 - 2, n6 - 2) = 4 * t3; BAND(n6 - 2, n6 - 1) = 5 * t4;
        BAND(n6 - 1, n6 - 4) = 2; BAND(n6 - 1, n6 - 3) = 6 * t1; BAND(n6 - 1, n6 - 2) = 12 * t2; BAND(n6 - 1, n6 - 1) = 20 * t3;
    }
    if (lane < 9) {                                    // b.row(r) = headPVA.col(r)^T, tail likewise (CPU.hpp:440-442, 497-499)
        rhs[lane] = dp.headPVA[b * 9 + lane];
        rhs[(n6 - 3) * 3 + lane] = dp.tailPVA[b * 9 + lane];
    }
    __syncthreads();

    // ---- factorizeLU (traj.hpp:655-687): per pivot, 36 lanes own the (row, column) pairs of the trailing block ----
    {
        const int a = lane / 6, bb = lane % 6;
        for (int k = 0; k <= n6 - 2; k++) {
            const int i = k + 1 + a, j = k + 1 + bb;
            const bool act = lane < 36 && i < n6 && j < n6;
            double l = 0.0, u = 0.0, v = 0.0;
            if (act) {
                l = BAND(i, k);
                u = BAND(k, j);
                v = BAND(i, j);
                if (l != 0.0) l /= BAND(k, k);
            }
            __syncthreads();
            if (act) {
                if (bb == 0) BAND(i, k) = l;
                if (u != 0.0 && l != 0.0) BAND(i, j) = v - l * u;
            }
            __syncthreads();
        }
    }
    // ---- solve (traj.hpp:692-719): 18 lanes = 6 rows below/above x 3 columns ----
    {
        const int a = lane / 3, c = lane % 3;
        for (int j = 0; j <= n6 - 1; j++) {
            const int i = j + 1 + a;
            if (lane < 18 && i < n6) {
                const double l = BAND(i, j);
                if (l != 0.0) rhs[i * 3 + c] -= l * rhs[j * 3 + c];
            }
            __syncthreads();
        }
        for (int j = n6 - 1; j >= 0; j--) {
            double xj = 0.0;
            if (lane < 18) xj = rhs[j * 3 + c] / BAND(j, j);
            __syncthreads();
            if (lane < 18) {
                if (a == 0) rhs[j * 3 + c] = xj;
                const int i = j - 1 - a;
                if (i >= 0) {
                    const double u = BAND(i, j);
                    if (u != 0.0) rhs[i * 3 + c] -= u * xj;
                }
            }
            __syncthreads();
        }
    }
    for (int i = lane; i < n6 * 3; i += 64) Cout[(size_t)p0 * 18 + i] = rhs[i];
    if (bandOut) {
        double *bo = bandOut + dp.boff[b];
        for (int i = lane; i < n6 * FRX_BAND_W; i += 64) bo[i] = band[i];
    }
}

// ---------------------------------------------------------------------------------------------
// k_penalty: ONE WAVE per workgroup; the wave owns ppw = 64/lpp consecutive pieces, lpp = min(kappa+1, 64)
// lanes per piece, lane = one quadrature sample (strided when kappa+1 > 64).
// Every global read of the wave is an independent, contiguous, coalesced sweep (coefficients, durations, and the pieces'
// corridor blocks hblk[gp] = {origin xyz, K | K x (unit normal, c)} padded to Kmax); the first trip of all three sweeps is in
// flight before the first LDS store (one memory latency, not three): no index-dependent second round of loads, no block-wide
// barrier (PMC on the 4-wave version: 61 % of wave cycles parked on s_waitcnt/s_barrier behind three dependent load rounds).
// Dynamic LDS (doubles): cS[ppw*18] | tS[ppw] | hS[ppw*(Kmax+1)*4] | red[64*21]
// ---------------------------------------------------------------------------------------------
// View of doubles in LDS for penalty_sample (frx_math.hpp): a 32-bit LDS address, so every read is a ds_read_b64 (pairs merge into
// ds_read2_b64) whatever the optimiser can or cannot prove about the pointer's address space, and fence() makes the address opaque:
// what was read before it is not kept in registers across a phase boundary but read again.
typedef const __attribute__((address_space(3))) double *lds_cdptr;
#ifndef FRX_PEN_OCC
#define FRX_PEN_OCC 3
#endif
struct LdsView {
    unsigned off;
    __device__ __forceinline__ explicit LdsView(const double *p) : off((unsigned)(uintptr_t)(lds_cdptr)p) {}
    __device__ __forceinline__ double operator[](int i) const { return ((lds_cdptr)(uintptr_t)off)[i]; }
    __device__ __forceinline__ void fence() { asm volatile("" : "+v"(off)); }
};

// The 20 partials {cost, d/dT, d/dc[6][3]} of ONE quadrature sample (sample j of its piece, local time s1 = step j, trapezoid weight omg).
template <bool LAT>
__device__ __forceinline__ void penalty_sample_partials(const DevProblem &dp, LdsView &c, LdsView &hb, int K, int Kmax, double s1, double step, double omg, double invK, int j, double (&o)[20]) {
    double adj[12], Ps, gTa;
    penalty_sample<LAT>(c, s1, omg * step, dp.pc, hb, K, Kmax, adj, Ps, gTa);
    if (!LAT) { c.fence(); FRX_PHASE(); }
    o[0] = omg * step * Ps; o[1] = (invK * j) * gTa + omg * Ps * invK;   // CPU.hpp:259,342-343
    const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
    const double b0[6] = {1.0, s1, s2, s3, s4, s5};
    const double b1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
    const double b2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
    const double b3[6] = {0.0, 0.0, 0.0, 6.0, 24.0 * s1, 60.0 * s2};
    // beta_m (x) a_m, m = 0..3, WITHOUT the structural zeros of the derivative bases (beta_m[k] = 0 for k < m).  Written as the full 4-term sum the
    // compiler kept them - it may not fold 0 * x, which is NaN for a non-finite x: 18 of the 72 FMAs of a sample multiplied by a literal zero
    // (v_fmac_f64 v, 0, v in the round-4 ISA).  Same products, same order of the remaining additions: the finite results are bit for bit the old ones.
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
        for (int d = 0; d < 3; d++) {
            double acc = b0[k] * adj[d];
            if (k >= 1) acc = acc + b1[k] * adj[3 + d];
            if (k >= 2) acc = acc + b2[k] * adj[6 + d];
            if (k >= 3) acc = acc + b3[k] * adj[9 + d];
            o[2 + 3 * k + d] = acc;
        }
}
// The samples of ONE lane (sample jl, jl + lpp, ... of the piece whose coefficients are at cS and corridor block at hS, duration Tp): their 20 partials
// {cost, d/dT, d/dc[6][3]} accumulated into the lane's LDS slot `mine`.
template <bool LAT>
__device__ __forceinline__ void penalty_lane_samples(const DevProblem &dp, const double *cS, const double *hS, double Tp, int jl, int lpp, int Kmax, double *mine) {
    LdsView c(cS), hb(hS);
    const int K = (int)hb[3];
    const int kappa = dp.kappa;
    const double step = Tp / kappa;                           // CPU.hpp:245
    const double invK = dp.inv_kappa;
    bool first = true;
    for (int j = jl; j <= kappa; j += lpp) {
        const double s1 = step * j;                           // sample abscissa as cc.cu:152
        const double omg = (j == 0 || j == kappa) ? 0.5 : 1.0;   // CPU.hpp:306
        double o[20];
        penalty_sample_partials<LAT>(dp, c, hb, K, Kmax, s1, step, omg, invK, j, o);
        // (the 20 partials of this sample; with more than one sample per lane (kappa + 1 > 64) the lane's LDS slot accumulates)
        if (!first) {
#pragma unroll
            for (int i = 0; i < 20; i++) o[i] += mine[i];
        }
#pragma unroll
        for (int i = 0; i < 20; i++) mine[i] = o[i];
        first = false;
    }
}
// fixed-order reduction over the samples of each piece (lane slots red[lane * 21 ..]): thread = (piece of the group, value)
template <bool SH>
__device__ __forceinline__ void penalty_reduce(const double *red, int npieces, int lpp, double *__restrict__ out, int lane, int nthr, bool wt, ll_u64 *out_ll = nullptr, unsigned ll_tag = 0) {
    for (int idx = lane; idx < npieces * 20; idx += nthr) {
        const int p2 = idx / 20, v = idx - p2 * 20;
        const double *src = red + (p2 * lpp) * 21 + v;
        double s = 0.0;
        int l = 0;
        // the additions are one dependent chain in sample order (determinism); the LDS reads are not - sixteen in flight per block
        // (four at a time cost an LDS latency per four additions: five of them for the 17 samples of kappa = 16)
        for (; l + 16 <= lpp; l += 16) {
            double b[16];
#pragma unroll
            for (int j = 0; j < 16; j++) b[j] = src[(l + j) * 21];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; j++) s += b[j];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll 4
        for (; l < lpp; l++) s += src[l * 21];
        if (SH && out_ll) rk_ll_put(out_ll + 2 * idx, s, ll_tag, wt);      // resident caller: the partials travel as granules that the adjoint polls (no drain, no arrival count in front of it)
        else stg<SH>(out + idx, s, wt);
    }
}

// The body works on the pieces [gp0, gp0 + npieces) with a GROUP of nthr = 64 W threads (`lane` = index in the group) and `sm` = the
// group's private LDS.  W = 1: one wave (the resident round kernel gives every wave of a workgroup its own pieces and LDS); W > 1: the
// whole workgroup (the stage kernel).  Lane L owns sample L % lpp of piece L / lpp, so a group covers floor(64 W / lpp) whole pieces:
// with kappa + 1 = 17 samples per piece one wave uses 51 of its 64 lanes (49 at the stock kappa = 48), four waves 255 of 256 (245) -
// round 2 ran the stage kernel with W = 1 and left a fifth of every FP64 issue slot empty.  Every wave of the workgroup has to call it
// (it contains workgroup barriers), idle ones with npieces = 0.
template <bool SH, bool LAT = false>
__device__ __forceinline__ void penalty_body(const DevProblem &dp, const double *__restrict__ T, const double *__restrict__ C,
                                             double *__restrict__ out20, int lpp, int ppw, int Kmax, int gp0, int npieces, double *sm, int lane, bool wt = true,
                                             int nthr = 64, ll_u64 *out20ll = nullptr, unsigned ll_tag = 0) {
    const int hstride = (Kmax + 1) * 4;
    double *cS = sm;
    double *tS = cS + ppw * 18;
    double *hS = tS + ppw;
    double *red = hS + (size_t)ppw * hstride;
    const int pl = lane / lpp, jl = lane - pl * lpp;
    const int pfl = (dp.piece_active && pl < npieces) ? dp.piece_active[gp0 + pl] : DV_EVAL;   // issued with the sweeps below

    {
        const double *csrc = C + (size_t)gp0 * 18, *hsrc = dp.hblk + (size_t)gp0 * hstride;
        const int nc = npieces * 18, nh = npieces * hstride;
        // 16-byte loads (corridor blocks and coefficient blocks are 16-byte multiples and 16-byte aligned in HBM); the LDS side is written
        // as two 8-byte stores because a wave's private LDS region may start on an odd double.  Trip 0 of every sweep first, into registers.
        const double2 *h2 = (const double2 *)hsrc;
        const int nh2 = nh >> 1, nc2 = nc >> 1;
        double2 hv0 = make_double2(0.0, 0.0), hv1 = hv0, cv0 = hv0;
        double ca = 0.0, cb = 0.0, tv = 0.0;
        if (lane < nh2) hv0 = h2[lane];
        if (lane + nthr < nh2) hv1 = h2[lane + nthr];
        if (SH) {                                                   // (C, T) come from another workgroup of the same launch: 8-byte L1-bypassing loads
            if (lane < nc) ca = ldg<SH>(csrc + lane);
            if (lane + nthr < nc) cb = ldg<SH>(csrc + lane + nthr);
        } else if (lane < nc2) cv0 = ((const double2 *)csrc)[lane];
        if (lane < npieces) tv = ldg<SH>(T + gp0 + lane);
        if (lane < nh2) { hS[2 * lane] = hv0.x; hS[2 * lane + 1] = hv0.y; }
        if (lane + nthr < nh2) { hS[2 * (lane + nthr)] = hv1.x; hS[2 * (lane + nthr) + 1] = hv1.y; }
        if (SH) {
            if (lane < nc) cS[lane] = ca;
            if (lane + nthr < nc) cS[lane + nthr] = cb;
            for (int i = lane + 2 * nthr; i < nc; i += nthr) cS[i] = ldg<SH>(csrc + i);
        } else {
            if (lane < nc2) { cS[2 * lane] = cv0.x; cS[2 * lane + 1] = cv0.y; }
            const double2 *c2p = (const double2 *)csrc;
            for (int i = lane + nthr; i < nc2; i += nthr) { const double2 v = c2p[i]; cS[2 * i] = v.x; cS[2 * i + 1] = v.y; }
        }
        if (lane < npieces) tS[lane] = tv;
#pragma unroll 2
        for (int i = lane + 2 * nthr; i < nh2; i += nthr) { const double2 v = h2[i]; hS[2 * i] = v.x; hS[2 * i + 1] = v.y; }      // corridor blocks beyond two trips
    }
    __syncthreads();

    const bool active = pl < npieces && (pfl & DV_EVAL);
    if (active) penalty_lane_samples<LAT>(dp, cS + pl * 18, hS + (size_t)pl * hstride, tS[pl], jl, lpp, Kmax, red + lane * 21);
    __syncthreads();
    penalty_reduce<SH>(red, npieces, lpp, out20 + (size_t)gp0 * 20, lane, nthr, wt, out20ll ? out20ll + (size_t)gp0 * 40 : nullptr, ll_tag);
}
// One WAVE on its own (the one-launch evaluation's members, frx_eval_kernel.hpp): the pieces [gp0, gp0 + npieces) of a wave-task with `sm` = the wave's private LDS.
// Nothing of the workgroup is involved - no barrier, the waves of a member run apart - and (C, T) are not loaded but POLLED: they arrive as granules
// {coefficient q of piece p at ct_ll[2 (19 p + q)], duration at q = 18} tagged `tag` (GranuleOut, forward_knot_body), one or a few per lane, behind a gate
// word that the leader sets right behind them (tag << 4 | its XCD + 1; until then the wave loads its corridor blocks and sleeps between polls of the gate).  Same samples, same
// sums as penalty_body.  Returns false when a wait expired (status receives the code).
template <bool LAT>
__device__ __forceinline__ bool penalty_wave_ll(const DevProblem &dp, const ll_u64 *__restrict__ ct_ll, const unsigned *gate, unsigned tag, ll_u64 *__restrict__ out20ll,
                                                int lpp, int ppw, int Kmax, int gp0, int npieces, double *sm, int lane, unsigned *status, ll_u64 spin_ticks, unsigned my_xcc, long long *stamps = nullptr) {
#define PW_STAMP(slot) do { if (stamps && lane == 0) stamps[slot] = (long long)wall_clock64(); } while (0)   // (the 100 MHz counter all workgroups share: the shader clocks of two XCDs are unrelated)
    PW_STAMP(44);
    const int hstride = (Kmax + 1) * 4;
    double *cS = sm;
    double *tS = cS + ppw * 18;
    double *hS = tS + ppw;
    double *red = hS + (size_t)ppw * hstride;
    const int pl = lane / lpp, jl = lane - pl * lpp;
    {   // corridor blocks: constant, fetched while the leader still runs its forward map
        const double2 *h2 = (const double2 *)(dp.hblk + (size_t)gp0 * hstride);
        const int nh2 = (npieces * hstride) >> 1;
        for (int i0 = lane; i0 < nh2; i0 += 4 * 64) {
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + 64 * u; v[u] = h2[i < nh2 ? i : nh2 - 1]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + 64 * u; if (i < nh2) { hS[2 * i] = v[u].x; hS[2 * i + 1] = v[u].y; } }
        }
    }
    const ll_u64 t_end = (ll_u64)wall_clock64() + spin_ticks;
    bool ok = true;
    unsigned gv = 0;
    for (unsigned spins = 0;; spins++) {                                   // the gate: one word per cluster (every lane reads the same address: one request per wave); tag << 4 | the leader's XCD + 1
        gv = __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((gv >> 4) == tag) break;
        if ((spins & 31u) == 31u && (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || (ll_u64)wall_clock64() > t_end)) { ok = false; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    PW_STAMP(45);
    const int ng = npieces * 19;
    const ll_u64 *src = ct_ll + 2 * (size_t)gp0 * 19;
    typedef unsigned ll_v4u __attribute__((ext_vector_type(4)));
    for (int j0 = lane; ok && j0 - lane < ng; j0 += 3 * 64) {               // three granules per lane and trip, ONE 16-byte L1-bypassing load each
        const int ja = j0 < ng ? j0 : ng - 1, jb = j0 + 64 < ng ? j0 + 64 : ja, jc = j0 + 128 < ng ? j0 + 128 : ja;
        const bool more = j0 - lane + 64 < ng;                               // (wave-uniform: kappa >= 16 has at most 57 granules per wave-task - one load per poll)
        ll_v4u ra, rb, rc;
        for (unsigned spins = 0;; spins++) {
            if (more) asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                                   : "=&v"(ra), "=&v"(rb), "=&v"(rc) : "v"(src + 2 * ja), "v"(src + 2 * jb), "v"(src + 2 * jc) : "memory");
            else { asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ra) : "v"(src + 2 * ja) : "memory"); rb = ra; rc = ra; }
            if (ra.y == tag && ra.w == tag && rb.y == tag && rb.w == tag && rc.y == tag && rc.w == tag) break;
            if ((spins & 31u) == 31u && (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || (ll_u64)wall_clock64() > t_end)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        auto put = [&](int j, const ll_v4u &r) {
            const int p = j / 19, q = j - 19 * p;
            const double v = __longlong_as_double((long long)((ll_u64)r.x | ((ll_u64)r.z << 32)));
            if (q < 18) cS[p * 18 + q] = v; else tS[p] = v;
        };
        if (j0 < ng) put(ja, ra);
        if (j0 + 64 < ng) put(jb, rb);
        if (j0 + 128 < ng) put(jc, rc);
    }
    ok = __builtin_amdgcn_ballot_w64(!ok) == 0ull;                          // (wave-uniform)
    if (!ok) {
        unsigned expect = 0u;
        if (lane == 0) __hip_atomic_compare_exchange_strong(status, &expect, 3u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // the wave's LDS stores are in order with its LDS loads; nothing may be hoisted over them
    PW_STAMP(46);
    if (pl < npieces) penalty_lane_samples<LAT>(dp, cS + pl * 18, hS + (size_t)pl * hstride, tS[pl], jl, lpp, Kmax, red + lane * 21);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PW_STAMP(47);
    const bool wt = my_xcc == 0u || (gv & 15u) != my_xcc;                  // the partials leave as plain stores when the leader runs on this XCD (its L2 is the meeting point), write-through otherwise
    penalty_reduce<true>(red, npieces, lpp, nullptr, lane, 64, wt, out20ll + (size_t)gp0 * 40, tag);
    PW_STAMP(48);
#undef PW_STAMP
    return true;
}
// Stage kernels: a workgroup of blockDim.x = 64 W threads owns ppg = floor(64 W / lpp) consecutive pieces (LaunchGeom::pen_w, ::ppg).
FRX_KERNEL_LINKAGE __global__ __launch_bounds__(256, 3) void k_penalty(DevProblem dp, const double *__restrict__ T, const double *__restrict__ C,
                                                    double *__restrict__ out20, int lpp, int ppg, int Kmax) {
    extern __shared__ double sm[];
    const int gp0 = blockIdx.x * ppg;
    penalty_body<false>(dp, T, C, out20, lpp, ppg, Kmax, gp0, min(ppg, dp.P - gp0), sm, threadIdx.x, true, blockDim.x);
}
// The latency form of the same kernel (penalty_sample<LAT>: no phase boundaries, 148 VGPRs) - the default: faster than the phased form at
// every batch size measured.
FRX_KERNEL_LINKAGE __global__ __launch_bounds__(256, FRX_PEN_OCC) void k_penalty_lat(DevProblem dp, const double *__restrict__ T, const double *__restrict__ C,
                                                        double *__restrict__ out20, int lpp, int ppg, int Kmax) {
    extern __shared__ double sm[];
    const int gp0 = blockIdx.x * ppg;
    penalty_body<false, true>(dp, T, C, out20, lpp, ppg, Kmax, gp0, min(ppg, dp.P - gp0), sm, threadIdx.x, true, blockDim.x);
}

// Large batches, one sample per lane (kappa + 1 <= 64), four-wave workgroups: the SAME samples and the same fixed-order sums, but the 20 partials cross the
// LDS transpose in TWO halves of ten, so that the transpose buffer is [256][11] instead of [256][21] doubles - 29 instead of 49 KB per workgroup at
// K = 8 - and the kernel is compiled for four waves per SIMD (128 VGPRs).  The one-phase form fits three workgroups on a CU by LDS and three waves on a SIMD by
// registers, and its launch at 1024 candidates spends a quarter of its wave cycles waiting: more workgroups in flight hide the staging trips of one
// another.  Bit-identical partials (each value is still summed over its piece's samples in sample order); FRX_PENALTY_TWOPHASE=0 keeps the one-phase launch.
// Dynamic LDS (doubles): cS[ppg*18] | tS[ppg] | hS[ppg*(Kmax+1)*4] | red[256 * 11]
//
// Round 6 - the instruction diet (520 VALU instructions per wave of which 298 were FP64 arithmetic).  What the other 222 were, from the ISA,
// and where they went:
//   * LPP (lanes per piece = kappa + 1) is a TEMPLATE parameter for the two resolutions the boundary is quoted on (17: BASELINE configs[2..4]; 49: the stock kappa = 48
//     of configs[0]); 0 = any other, at run time.  `lane / lpp` was a run-time integer division (a v_rcp_iflag_f32 sequence of 25 instructions), the trip count of the
//     `#pragma unroll 2` staging loop - a division by the run-time blockDim.x - another one, and the fixed-order reduction a three-level loop nest (16 reads in flight,
//     remainder, unroll 4) with its address arithmetic; with LPP known the reduction is 17 (or 16 + 16 + 17) reads at immediate offsets and their additions.
//   * the workgroup is 256 threads by definition of this form (LaunchGeom::pen_w == 4): a constant, not blockDim.x.
//   * step = T / kappa (an IEEE division: 25 instructions) was computed by every lane of every wave; the few lanes that stage the durations now divide once per piece and
//     leave the STEP in LDS (the duration itself is not used by a sample) - one wave of four pays.
//   * `double o[20] = {0}` for every lane in front of the sample (20 v_mov_b64 and as many phi copies behind the branch): lanes that own no sample store nothing - their
//     slots are never read - and lanes of a piece that is switched off (piece_active) store zeros in a branch of their own.
template <int LPP>
__global__ __launch_bounds__(256, 4) void k_penalty_lat2(DevProblem dp, const double *__restrict__ T, const double *__restrict__ C,
                                                         double *__restrict__ out20, int lpp_rt, int ppg, int Kmax) {
    extern __shared__ double sm[];
    constexpr int nthr = 256;
    const int lpp = LPP ? LPP : lpp_rt;
    const int lane = threadIdx.x;
    const int gp0 = blockIdx.x * ppg, npieces = min(ppg, dp.P - gp0);
    const int hstride = (Kmax + 1) * 4;
    double *cS = sm, *tS = cS + ppg * 18, *hS = tS + ppg, *red = hS + (size_t)ppg * hstride;
    const int pl = lane / lpp, jl = lane - pl * lpp;
    const bool exists = pl < npieces;
    const int pfl = (dp.piece_active && exists) ? dp.piece_active[gp0 + pl] : DV_EVAL;
    {   // staging (16-byte loads, every sweep's first trips in flight before the first LDS store); corridor blocks: two trips cover K <= 16 at kappa = 16
        const double2 *h2 = (const double2 *)(dp.hblk + (size_t)gp0 * hstride), *c2 = (const double2 *)(C + (size_t)gp0 * 18);
        const int nh2 = (npieces * hstride) >> 1, nc2 = (npieces * 18) >> 1;
        double2 hv0, hv1, cv0;
        double tv;
        const bool h0 = lane < nh2, h1 = lane + nthr < nh2, c0 = lane < nc2, t0 = lane < npieces;
        if (h0) hv0 = h2[lane];
        if (h1) hv1 = h2[lane + nthr];
        if (c0) cv0 = c2[lane];
        if (t0) tv = T[gp0 + lane];
        if (h0) { hS[2 * lane] = hv0.x; hS[2 * lane + 1] = hv0.y; }
        if (h1) { hS[2 * (lane + nthr)] = hv1.x; hS[2 * (lane + nthr) + 1] = hv1.y; }
        if (c0) { cS[2 * lane] = cv0.x; cS[2 * lane + 1] = cv0.y; }
#pragma unroll 1
        for (int i = lane + nthr; i < nc2; i += nthr) { const double2 v = c2[i]; cS[2 * i] = v.x; cS[2 * i + 1] = v.y; }      // (more than 28 pieces per workgroup: kappa < 9)
        if (t0) tS[lane] = tv / dp.kappa;                                      // the piece's STEP, CPU.hpp:245 - once per piece
#pragma unroll 1
        for (int i = lane + 2 * nthr; i < nh2; i += nthr) { const double2 v = h2[i]; hS[2 * i] = v.x; hS[2 * i + 1] = v.y; }
    }
    __syncthreads();
    const bool active = exists && (pfl & DV_EVAL);
    double o[20];
    if (active) {
        LdsView c(cS + pl * 18), hb(hS + (size_t)pl * hstride);
        const int K = (int)hb[3], kappa = dp.kappa;
        const double step = tS[pl];
        penalty_sample_partials<true>(dp, c, hb, K, Kmax, step * jl, step, (jl == 0 || jl == kappa) ? 0.5 : 1.0, dp.inv_kappa, jl, o);
    }
    double *mine = red + lane * 11;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (half) __syncthreads();                                            // the first half has been summed: its slots are free
        if (active) {
#pragma unroll
            for (int i = 0; i < 10; i++) mine[i] = o[10 * half + i];
        } else if (exists) {                                                  // a piece that is switched off this round: its partials are zeros (nobody reads the slots of lanes without a piece)
#pragma unroll
            for (int i = 0; i < 10; i++) mine[i] = 0.0;
        }
        __syncthreads();
#pragma unroll 1
        for (int idx = lane; idx < npieces * 10; idx += nthr) {               // (one trip unless a workgroup owns more than 25 pieces)
            const int p2 = idx / 10, v = idx - p2 * 10;
            const double *src = red + (p2 * lpp) * 11 + v;
            double s = 0.0;
            if (LPP) {
                // the additions are one dependent chain in sample order (determinism); the LDS reads are not - up to seventeen in flight per block
                constexpr int BLK = LPP <= 24 ? (LPP ? LPP : 1) : 16;
                int l = 0;
#pragma unroll
                for (; l + BLK <= (LPP ? LPP : 1); l += BLK) {
                    double b[BLK];
#pragma unroll
                    for (int j = 0; j < BLK; j++) b[j] = src[(l + j) * 11];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < BLK; j++) s += b[j];
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (; l < LPP; l++) s += src[l * 11];
            } else {
                int l = 0;
                for (; l + 16 <= lpp; l += 16) {                              // (as penalty_reduce: sixteen reads in flight, one chain of additions in sample order)
                    double b[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) b[j] = src[(l + j) * 11];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 16; j++) s += b[j];
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll 4
                for (; l < lpp; l++) s += src[l * 11];
            }
            out20[(size_t)(gp0 + p2) * 20 + 10 * half + v] = s;
        }
    }
}

// The round-5 form of the same kernel, kept for build-against-build measurements in ONE process (FRX_PENALTY_TWOPHASE=5; scripts/r06/penalty_ab.py): run-time lpp and
// blockDim.x, the step divided by every lane, zero-initialised partials.  Bit-identical output.
FRX_KERNEL_LINKAGE __global__ __launch_bounds__(256, 4) void k_penalty_lat2_r5(DevProblem dp, const double *__restrict__ T, const double *__restrict__ C,
                                                         double *__restrict__ out20, int lpp, int ppg, int Kmax) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x, nthr = blockDim.x;
    const int gp0 = blockIdx.x * ppg, npieces = min(ppg, dp.P - gp0);
    const int hstride = (Kmax + 1) * 4;
    double *cS = sm, *tS = cS + ppg * 18, *hS = tS + ppg, *red = hS + (size_t)ppg * hstride;
    const int pl = lane / lpp, jl = lane - pl * lpp;
    const int pfl = (dp.piece_active && pl < npieces) ? dp.piece_active[gp0 + pl] : DV_EVAL;
    {   // staging as in penalty_body (16-byte loads, every sweep's first trips in flight before the first LDS store)
        const double2 *h2 = (const double2 *)(dp.hblk + (size_t)gp0 * hstride), *c2 = (const double2 *)(C + (size_t)gp0 * 18);
        const int nh2 = (npieces * hstride) >> 1, nc2 = (npieces * 18) >> 1;
        double2 hv0 = make_double2(0.0, 0.0), hv1 = hv0, cv0 = hv0;
        double tv = 0.0;
        if (lane < nh2) hv0 = h2[lane];
        if (lane + nthr < nh2) hv1 = h2[lane + nthr];
        if (lane < nc2) cv0 = c2[lane];
        if (lane < npieces) tv = T[gp0 + lane];
        if (lane < nh2) { hS[2 * lane] = hv0.x; hS[2 * lane + 1] = hv0.y; }
        if (lane + nthr < nh2) { hS[2 * (lane + nthr)] = hv1.x; hS[2 * (lane + nthr) + 1] = hv1.y; }
        if (lane < nc2) { cS[2 * lane] = cv0.x; cS[2 * lane + 1] = cv0.y; }
        for (int i = lane + nthr; i < nc2; i += nthr) { const double2 v = c2[i]; cS[2 * i] = v.x; cS[2 * i + 1] = v.y; }
        if (lane < npieces) tS[lane] = tv;
#pragma unroll 2
        for (int i = lane + 2 * nthr; i < nh2; i += nthr) { const double2 v = h2[i]; hS[2 * i] = v.x; hS[2 * i + 1] = v.y; }
    }
    __syncthreads();
    const bool active = pl < npieces && (pfl & DV_EVAL);
    double o[20];
#pragma unroll
    for (int i = 0; i < 20; i++) o[i] = 0.0;
    if (active) {
        LdsView c(cS + pl * 18), hb(hS + (size_t)pl * hstride);
        const int K = (int)hb[3], kappa = dp.kappa;
        const double step = tS[pl] / kappa;                                   // CPU.hpp:245
        penalty_sample_partials<true>(dp, c, hb, K, Kmax, step * jl, step, (jl == 0 || jl == kappa) ? 0.5 : 1.0, dp.inv_kappa, jl, o);
    }
    double *mine = red + lane * 11;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (half) __syncthreads();                                            // the first half has been summed: its slots are free
#pragma unroll
        for (int i = 0; i < 10; i++) mine[i] = o[10 * half + i];
        __syncthreads();
        for (int idx = lane; idx < npieces * 10; idx += nthr) {
            const int p2 = idx / 10, v = idx - p2 * 10;
            const double *src = red + (p2 * lpp) * 11 + v;
            double s = 0.0;
            int l = 0;
            for (; l + 16 <= lpp; l += 16) {                                  // (as penalty_reduce: sixteen reads in flight, one chain of additions in sample order)
                double b[16];
#pragma unroll
                for (int j = 0; j < 16; j++) b[j] = src[(l + j) * 11];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 16; j++) s += b[j];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll 4
            for (; l < lpp; l++) s += src[l * 11];
            out20[(size_t)(gp0 + p2) * 20 + 10 * half + v] = s;
        }
    }
}

// (Streaming forms for large batches - persistent workgroups walking over the groups of pieces with the next group's operands in flight during the
// current group's samples - were built twice and measured slower than one group per workgroup both times: with the operands prefetched into
// registers (round 2: 43.9 vs 40.2 us at 1024 candidates; the prefetch forces the phased sample) and with global_load_lds_dwordx4 into a second
// LDS buffer, no registers involved (round 3: bit-equal, 44.9-55.2 vs 36.3 us at 1024 candidates, 166-210 vs 134 us at 4096,
// profiles/r03_penalty_stream.jsonl).  Operand latency is not what the launch waits for: DESIGN.md 3.2.)

// ---------------------------------------------------------------------------------------------
// k_backward: grid = B, block = 64.  Dynamic LDS: band[6N*13] | gd[6N*3] | cL[6N*3] | Tf[N] | gT[N] | gC[cN]
// ---------------------------------------------------------------------------------------------
FRX_KERNEL_LINKAGE __global__ __launch_bounds__(64) void k_backward(DevProblem dp, const double *__restrict__ x, const double *__restrict__ Tin,
                                                 const double *__restrict__ Cin, const double *__restrict__ bandIn,
                                                 const double *__restrict__ out20, double *__restrict__ f, double *__restrict__ g,
                                                 int maxN, int maxCN) {
    extern __shared__ double sm[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int p0 = dp.poff[b], N = dp.poff[b + 1] - p0, n6 = 6 * N;
    const int c0 = dp.coff[b], cN = dp.coff[b + 1] - c0;
    const int x0 = dp.xoff[b];
    double *band = sm;
    double *gd = band + (size_t)6 * maxN * FRX_BAND_W;
    double *cL = gd + (size_t)6 * maxN * 3;
    double *Tf = cL + (size_t)6 * maxN * 3;
    double *gT = Tf + maxN;
    double *gCo = gT + maxN;

    {
        const double *bi = bandIn + dp.boff[b];
        for (int i = lane; i < n6 * FRX_BAND_W; i += 64) band[i] = bi[i];
        for (int i = lane; i < n6 * 3; i += 64) cL[i] = Cin[(size_t)p0 * 18 + i];
        for (int i = lane; i < N; i += 64) Tf[i] = Tin[p0 + i];
    }
    __syncthreads();

    // ---- jerk cost + gradients (CPU.hpp:507-520, 65-95) added to the penalty partials ----
    double costAcc = 0.0;
    for (int i = lane; i < N; i += 64) {
        const double t1 = Tf[i], t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
        const double *c3 = cL + (6 * i + 3) * 3, *c4 = cL + (6 * i + 4) * 3, *c5 = cL + (6 * i + 5) * 3;
        const double s33 = dot3(c3, c3), s43 = dot3(c4, c3), s44 = dot3(c4, c4), s53 = dot3(c5, c3), s54 = dot3(c5, c4), s55 = dot3(c5, c5);
        const double *o = out20 + (size_t)(p0 + i) * 20;
        costAcc += o[0] + (36.0 * s33 * t1 + 144.0 * s43 * t2 + 192.0 * s44 * t3 + 240.0 * s53 * t3 + 720.0 * s54 * t4 + 720.0 * s55 * t5);
        gT[i] = o[1] + (36.0 * s33 + 288.0 * s43 * t1 + 576.0 * s44 * t2 + 720.0 * s53 * t2 + 2880.0 * s54 * t3 + 3600.0 * s55 * t4);
        double *gi = gd + (size_t)i * 18;
#pragma unroll
        for (int v = 0; v < 9; v++) gi[v] = o[2 + v];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            gi[9 + d] = o[2 + 9 + d] + (72.0 * c3[d] * t1 + 144.0 * c4[d] * t2 + 240.0 * c5[d] * t3);
            gi[12 + d] = o[2 + 12 + d] + (144.0 * c3[d] * t2 + 384.0 * c4[d] * t3 + 720.0 * c5[d] * t4);
            gi[15 + d] = o[2 + 15 + d] + (240.0 * c3[d] * t3 + 720.0 * c4[d] * t4 + 1440.0 * c5[d] * t5);
        }
    }
    __syncthreads();

    // ---- solveAdj (traj.hpp:724-751) ----
    {
        const int a = lane / 3, c = lane % 3;
        for (int j = 0; j <= n6 - 1; j++) {
            double xj = 0.0;
            if (lane < 18) xj = gd[j * 3 + c] / BAND(j, j);
            __syncthreads();
            if (lane < 18) {
                if (a == 0) gd[j * 3 + c] = xj;
                const int i = j + 1 + a;
                if (i < n6) {
                    const double u = BAND(j, i);
                    if (u != 0.0) gd[i * 3 + c] -= u * xj;
                }
            }
            __syncthreads();
        }
        for (int j = n6 - 1; j >= 0; j--) {
            const int i = j - 1 - a;
            if (lane < 18 && i >= 0) {
                const double l = BAND(j, i);
                if (l != 0.0) gd[i * 3 + c] -= l * gd[j * 3 + c];
            }
            __syncthreads();
        }
    }

    // ---- addPropCtoT (CPU.hpp:104-151) ----
    for (int i = lane; i < N; i += 64) {
        const double t1 = Tf[i], t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2;
        const double *c1 = cL + (6 * i + 1) * 3, *c2 = c1 + 3, *c3 = c1 + 6, *c4 = c1 + 9, *c5 = c1 + 12;
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double negVel = -(c1[d] + 2.0 * t1 * c2[d] + 3.0 * t2 * c3[d] + 4.0 * t3 * c4[d] + 5.0 * t4 * c5[d]);
            const double negAcc = -(2.0 * c2[d] + 6.0 * t1 * c3[d] + 12.0 * t2 * c4[d] + 20.0 * t3 * c5[d]);
            const double negJer = -(6.0 * c3[d] + 24.0 * t1 * c4[d] + 60.0 * t2 * c5[d]);
            if (i < N - 1) {
                const double negSnp = -(24.0 * c4[d] + 120.0 * t1 * c5[d]);
                const double negCrk = -120.0 * c5[d];
                const double *ad = gd + (6 * i + 3) * 3 + d;     // rows 6i+3 .. 6i+8
                s += negSnp * ad[0] + negCrk * ad[3] + negVel * ad[6] + negVel * ad[9] + negAcc * ad[12] + negJer * ad[15];
            } else {
                const double *ad = gd + (n6 - 3) * 3 + d;
                s += negVel * ad[0] + negAcc * ad[3] + negJer * ad[6];
            }
        }
        gT[i] += s + dp.rho;                                     // + rho: CPU.hpp:989
    }
    __syncthreads();

    // ---- cost (CPU.hpp:988) and mergeToCoarseGradT (CPU.hpp:946-959) ----
    double sumTc = 0.0;
    for (int i = lane; i < cN; i += 64) {
        const int gc = c0 + i;
        const int iv = dp.coarse_iv[gc], fb = dp.coarse_fbeg[gc] - p0;
        double s = 0.0, tt = 0.0;
        for (int a = 0; a < iv; a++) { s += gT[fb + a]; tt += Tf[fb + a]; }
        gCo[i] = s / iv;
        sumTc += tt;
    }
    const double total = wave_sum(costAcc) + dp.rho * wave_sum(sumTc);
    if (lane == 0) f[b] = total;
    __syncthreads();

    // ---- addLayerTGrad (CPU.hpp:816-894) ----
    if (dp.soft) {
        for (int i = lane; i < cN; i += 64) g[x0 + i] = gCo[i] * dT_dtau(x[x0 + i], dp.c2 != 0);
    } else if (lane == 0) {
        const int Ms1 = cN - 1;
        const double gTail = dp.sumT * gCo[Ms1];
        double expTauSum = 0.0, gFreeDotExpTau = 0.0;
        for (int i = 0; i < Ms1; i++) {
            const double e = tau_to_T(x[x0 + i], dp.c2 != 0);
            expTauSum += e;
            gFreeDotExpTau += e * (dp.sumT * gCo[i]);
        }
        const double den = expTauSum + 1.0;
        for (int i = 0; i < Ms1; i++) {
            const double de = dT_dtau(x[x0 + i], dp.c2 != 0);
            g[x0 + i] = (dp.sumT * gCo[i] - gTail) * de / den - (gFreeDotExpTau - gTail * expTauSum) * de / (den * den);
        }
    }
    // ---- addPropCtoP + addLayerPGrad (CPU.hpp:154-161, 897-928) ----
    const int w0 = p0 - b;
    for (int i = lane; i < N - 1; i += 64) {
        const int gw = w0 + i;
        const int k = dp.wp_nv[gw] - 1;
        const double *V = dp.vrec + 3 * (size_t)dp.wp_vbeg[gw];
        const int xb = dp.wp_xbeg[gw];
        const double *xi = x + xb;
        const double gq0 = gd[(6 * i + 5) * 3], gq1 = gd[(6 * i + 5) * 3 + 1], gq2 = gd[(6 * i + 5) * 3 + 2];
        double qn = 0.0;
        for (int a = 0; a < k; a++) qn += xi[a] * xi[a];
        const double qp1 = qn + 1.0, qp1sq = qp1 * qp1, sc = 2.0 / qp1;
        double gdq = 0.0;
        for (int a = 0; a < k; a++) {
            const double r = sc * xi[a];
            const double gdr = (V[3 * (a + 1)] * gq0 + V[3 * (a + 1) + 1] * gq1 + V[3 * (a + 1) + 2] * gq2) * r * 2.0;
            gdq += gdr * xi[a];
        }
        for (int a = 0; a < k; a++) {
            const double r = sc * xi[a];
            const double gdr = (V[3 * (a + 1)] * gq0 + V[3 * (a + 1) + 1] * gq1 + V[3 * (a + 1) + 2] * gq2) * r * 2.0;
            g[xb + a] = gdr * 2.0 / qp1 - xi[a] * 4.0 * gdq / qp1sq;
        }
    }
}

// =============================================================================================
// Knot-form MINCO map (frx_minco.hpp): O(log N) depth instead of five 6N-row sequential sweeps.
// One workgroup per candidate, thread k = piece k (0..N-1) AND interior knot k (1..N-1).
// Block = 256 threads: all of them stage the candidate's x and waypoint polytopes (every global read issued up front),
// threads < nrow = 64*ceil(maxN/64) own a piece / knot.
// LDS (doubles): rows[2][18][nrow] (SoA, conflict-free) | KP,KV,KA [3][nrow+1] x 3 arrays | Tf[nrow] | Tc[maxCN] | red | xs | vs
// =============================================================================================
#define FRX_STAMP(slot) do { if (dp.stamps && b == 0 && k == 0) dp.stamps[slot] = (long long)__builtin_readcyclecounter(); } while (0)
// the same from the first lane of axis wave 1 (thread 64): the axis waves' own timeline in the wave-specialised bodies
#define FRX_STAMP_AX(slot) do { if (dp.stamps && b == 0 && k == 64) dp.stamps[slot] = (long long)__builtin_readcyclecounter(); } while (0)
// Row buffer: [2][9][nrow] double2 - a block row is 18 doubles = 9 pairs (Dinv: 0,1; L: 2,3; U: 4,5; r: 6,7,8), one 16-byte LDS access
// per pair, consecutive lanes consecutive addresses.  The buffer holds D^-1, not D: a neighbour only ever needs the inverse of this
// row's diagonal block (alpha = L D_lo^-1), so the owner inverts once instead of both neighbours inverting the same block.
#define ROW2(buf, f, t) ((double2 *)rowbuf)[((buf) * 9 + (f)) * nrow + (t)]

__device__ __forceinline__ void row_store(double *rowbuf, int nrow, int buf, int t, const KnotRow &R, const double *Dinv) {
    ROW2(buf, 0, t) = make_double2(Dinv[0], Dinv[1]); ROW2(buf, 1, t) = make_double2(Dinv[2], Dinv[3]);
    ROW2(buf, 2, t) = make_double2(R.L[0], R.L[1]); ROW2(buf, 3, t) = make_double2(R.L[2], R.L[3]);
    ROW2(buf, 4, t) = make_double2(R.U[0], R.U[1]); ROW2(buf, 5, t) = make_double2(R.U[2], R.U[3]);
    ROW2(buf, 6, t) = make_double2(R.r[0], R.r[1]); ROW2(buf, 7, t) = make_double2(R.r[2], R.r[3]); ROW2(buf, 8, t) = make_double2(R.r[4], R.r[5]);
}
// neighbour row (its D field is the INVERSE of its diagonal block); `in` false = beyond the ends: the identity row
__device__ __forceinline__ void row_load(const double *rowbuf, int nrow, int buf, int t, bool in, KnotRow &R) {
    double2 q[9];
#pragma unroll
    for (int f = 0; f < 9; f++) q[f] = ROW2(buf, f, t);
    R.D[0] = in ? q[0].x : 1.0; R.D[1] = in ? q[0].y : 0.0; R.D[2] = in ? q[1].x : 0.0; R.D[3] = in ? q[1].y : 1.0;
    R.L[0] = in ? q[2].x : 0.0; R.L[1] = in ? q[2].y : 0.0; R.L[2] = in ? q[3].x : 0.0; R.L[3] = in ? q[3].y : 0.0;
    R.U[0] = in ? q[4].x : 0.0; R.U[1] = in ? q[4].y : 0.0; R.U[2] = in ? q[5].x : 0.0; R.U[3] = in ? q[5].y : 0.0;
    R.r[0] = in ? q[6].x : 0.0; R.r[1] = in ? q[6].y : 0.0; R.r[2] = in ? q[7].x : 0.0; R.r[3] = in ? q[7].y : 0.0; R.r[4] = in ? q[8].x : 0.0; R.r[5] = in ? q[8].y : 0.0;
}

// Parallel cyclic reduction over knots 1..N-1 (thread k owns knot k); returns this knot's (v, a) solution.
// `save` (optional): per-step multipliers and the final D^-1 of this knot, AoS save[gk * sstride + step*8 + i] and
// save[gk * sstride + nsteps*8 + i] (sstride = nsteps*8 + 4), for k_backward_knot's right-hand-side-only reduction.
__device__ __forceinline__ void pcr_solve_wg(double *rowbuf, int nrow, int k, int N, KnotRow &me, double *v, double *a,
                                             double *save, size_t sstride, size_t gk, int nsteps) {
    const bool act = k >= 1 && k <= N - 1;
    int buf = 0, step = 0;
    double Dinv[4] = {1.0, 0.0, 0.0, 1.0};
    double2 *sp = save ? (double2 *)(save + gk * sstride) : nullptr;         // sstride is even: 16-byte aligned
    if (act) { m2_inv(me.D, Dinv); row_store(rowbuf, nrow, 0, k, me, Dinv); }
    __syncthreads();
    // Inside the loop the barrier only has to order LDS traffic.  __syncthreads() also waits for vmcnt(0), and on gfx9 the
    // multiplier stores below count in vmcnt: every step would wait for its stores to be acknowledged by L2.
    for (int s = 1; s < N - 1; s <<= 1, step++) {
        if (act) {
            KnotRow lo, hi, out;
            double al[4], be[4];
            const bool inlo = k - s >= 1, inhi = k + s <= N - 1;
            row_load(rowbuf, nrow, buf, inlo ? k - s : k, inlo, lo);      // clamped address, select afterwards: no divergent paths
            row_load(rowbuf, nrow, buf, inhi ? k + s : k, inhi, hi);
            pcr_step_inv(me, lo, hi, out, al, be);
            me = out;
            m2_inv(me.D, Dinv);
            row_store(rowbuf, nrow, buf ^ 1, k, me, Dinv);
            if (sp) {
                sp[step * 4 + 0] = make_double2(al[0], al[1]); sp[step * 4 + 1] = make_double2(al[2], al[3]);
                sp[step * 4 + 2] = make_double2(be[0], be[1]); sp[step * 4 + 3] = make_double2(be[2], be[3]);
            }
        }
        buf ^= 1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (act) {
#pragma unroll
        for (int x3 = 0; x3 < 3; x3++) {                            // decoupled row: w = D^-1 r (pcr_finish)
            v[x3] = Dinv[0] * me.r[x3] + Dinv[1] * me.r[3 + x3];
            a[x3] = Dinv[2] * me.r[x3] + Dinv[3] * me.r[3 + x3];
        }
        if (sp) { sp[nsteps * 4] = make_double2(Dinv[0], Dinv[1]); sp[nsteps * 4 + 1] = make_double2(Dinv[2], Dinv[3]); }
    }
}

// Right-hand-side-only reduction with the multipliers saved by the forward pass of the same evaluation (K is symmetric:
// the adjoint system is the same matrix).  r = (rv[3], ra[3]) in, solution out.  LDS: rbuf[2][6][nrow].
#define RBF(buf, f, t) rbuf[((buf) * 6 + (f)) * nrow + (t)]
// `pw` = LDS copy [knot][nsteps*8+4 (+1 pad: odd stride, conflict-free)] of this candidate's saved multipliers.
__device__ __forceinline__ void pcr_apply_wg(double *rbuf, int nrow, int k, int N, double *r, const double *pw, int nsteps) {
    const bool act = k >= 1 && k <= N - 1;
    if (act) {
#pragma unroll
        for (int i = 0; i < 6; i++) RBF(0, i, k) = r[i];
    }
    __syncthreads();
    int buf = 0, st = 0;
    for (int s = 1; s < N - 1; s <<= 1, st++) {
        {
            if (act) {
                double ab[8];
#pragma unroll
                for (int i = 0; i < 8; i++) ab[i] = pw[k * (nsteps * 8 + 5) + st * 8 + i];
                double rlo[6] = {0, 0, 0, 0, 0, 0}, rhi[6] = {0, 0, 0, 0, 0, 0};
                if (k - s >= 1) {
#pragma unroll
                    for (int i = 0; i < 6; i++) rlo[i] = RBF(buf, i, k - s);
                }
                if (k + s <= N - 1) {
#pragma unroll
                    for (int i = 0; i < 6; i++) rhi[i] = RBF(buf, i, k + s);
                }
                pcr_rhs_step(r, ab, ab + 4, rlo, rhi);
#pragma unroll
                for (int i = 0; i < 6; i++) RBF(buf ^ 1, i, k) = r[i];
            }
            buf ^= 1;
            __syncthreads();
        }
    }
    if (act) {
        double Dinv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) Dinv[i] = pw[k * (nsteps * 8 + 5) + nsteps * 8 + i];
#pragma unroll
        for (int x3 = 0; x3 < 3; x3++) {
            const double rv = r[x3], ra = r[3 + x3];
            r[x3] = Dinv[0] * rv + Dinv[1] * ra;
            r[3 + x3] = Dinv[2] * rv + Dinv[3] * ra;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised reductions.  A PCR step of pcr_solve_wg is ~180 dependent FP64 instructions issued by ONE wave while the other three
// wait at the barrier (measured ~1800 cycles per step: latency of dependent FP64 operations with nothing else to issue on that SIMD).
// In the wave-specialised forms wave 0 reduces the MATRIX only (lane = knot) and publishes the step's multipliers in LDS; waves 1-3 each
// carry ONE AXIS of the right-hand side (lane = knot).  Entry by entry the arithmetic is that of pcr_step_inv / pcr_rhs_step.
//   <= 64 pieces  : forward = pcr_matrix_wave64 (free-running matrix wave, no barrier per step) + the axis code in forward_knot_body;
//                   adjoint = backward_knot_wsp64 (multipliers re-used, neighbours by lane shifts); pcr_waves_wg = the adjoint solve alone
//   65..128 pieces: pcr_waves2_wg (two knots per lane, barrier per step, right-hand sides through LDS)
//   matrix rows : MR2(buf, f, k) = (Dinv, L[, U]) as double2 pairs, [2][6][nrow] double2 at rowbuf
//   rhs         : RS(buf, f, k),  f < 6  = first / second equation x axis, [2][6][nrow] doubles behind them (pcr_waves2_wg, pcr_waves_wg)
//   multipliers : pw[k * pws + step * 8 + i] (alpha 0-3, beta 4-7), final D^-1 at pw[k * pws + nsteps * 8 + i]
// ---------------------------------------------------------------------------------------------
#define MR2(buf, f, t) ((double2 *)rowbuf)[((buf) * 6 + (f)) * nrow + (t)]
#define RS(buf, f, t) rsb[((buf) * 6 + (f)) * nrow + (t)]
__device__ __forceinline__ void pcr_axis_step(double *rsb, int nrow, int kk, int ax, int N, int s, int st, const double *pw, int pws, double &r0,
                                              double &r1) {
    const bool act = kk >= 1 && kk <= N - 1;
    const int buf = st & 1;
    const bool inlo = act && kk - s >= 1, inhi = act && kk + s <= N - 1;
    const int kc = act ? kk : 1, klo = inlo ? kk - s : kc, khi = inhi ? kk + s : kc;
    double ab[8];
#pragma unroll
    for (int i = 0; i < 8; i++) ab[i] = pw[kc * pws + st * 8 + i];
    const double l0r = RS(buf, ax, klo), l1r = RS(buf, 3 + ax, klo), h0r = RS(buf, ax, khi), h1r = RS(buf, 3 + ax, khi);
    const double l0 = inlo ? l0r : 0.0, l1 = inlo ? l1r : 0.0, h0 = inhi ? h0r : 0.0, h1 = inhi ? h1r : 0.0;
    const double n0 = r0 - (ab[0] * l0 + ab[1] * l1) - (ab[4] * h0 + ab[5] * h1);                // pcr_rhs_step
    const double n1 = r1 - (ab[2] * l0 + ab[3] * l1) - (ab[6] * h0 + ab[7] * h1);
    r0 = n0; r1 = n1;
    if (act) { RS(buf ^ 1, ax, kk) = r0; RS(buf ^ 1, 3 + ax, kk) = r1; }
}
// Right-hand sides alone with the multipliers in pw (the adjoint solve of <= 64 pieces; all 256 threads call it): RS(0, ..) holds the
// right-hand side, the solution goes to KV / KA.  Lane = knot, so the neighbours k -+ s are lane shifts: the exchange goes through
// the LDS crossbar (ds_bpermute, no memory, no barrier) and the three axis waves run their steps without meeting anyone.
// (The forward reduction of the matrix is pcr_matrix_wave64 below.)
__device__ __forceinline__ void pcr_waves_wg(double *rowbuf, int nrow, int t, int N, double *pw, int pws, int nsteps, double *KV, double *KA) {
    double *rsb = rowbuf + (size_t)24 * nrow;
    const int wave = t >> 6, kk = t & 63, ax = wave - 1;
    const bool act = kk >= 1 && kk <= N - 1;
    const int kc = act ? kk : 1;
    int nst = 0;
    for (int s = 1; s < N - 1; s <<= 1) nst++;
    if (wave < 1) return;
    double r0 = RS(0, ax, kc), r1 = RS(0, 3 + ax, kc);
    for (int st = 0; st < nst; st++) {
        const int s = 1 << st;
        const bool inlo = act && kk - s >= 1, inhi = act && kk + s <= N - 1;
        double ab[8];
#pragma unroll
        for (int i = 0; i < 8; i++) ab[i] = pw[kc * pws + st * 8 + i];
        const double l0r = __shfl_up(r0, s, 64), l1r = __shfl_up(r1, s, 64), h0r = __shfl_down(r0, s, 64), h1r = __shfl_down(r1, s, 64);
        const double l0 = inlo ? l0r : 0.0, l1 = inlo ? l1r : 0.0, h0 = inhi ? h0r : 0.0, h1 = inhi ? h1r : 0.0;
        const double n0 = r0 - (ab[0] * l0 + ab[1] * l1) - (ab[4] * h0 + ab[5] * h1);    // pcr_rhs_step
        const double n1 = r1 - (ab[2] * l0 + ab[3] * l1) - (ab[6] * h0 + ab[7] * h1);
        r0 = n0; r1 = n1;
    }
    if (act) {
        const double *Di = pw + kk * pws + nsteps * 8;
        KV[ax * (nrow + 1) + kk] = Di[0] * r0 + Di[1] * r1;
        KA[ax * (nrow + 1) + kk] = Di[2] * r0 + Di[3] * r1;
    }
}

// ---------------------------------------------------------------------------------------------
// The same wave specialisation for 65 .. 128 pieces (nrow = 128): TWO knots per lane (kk = lane and lane + 64).  Wave 0 reduces the
// matrix rows of both its knots per step (two independent dependency chains: better issue utilisation than one), waves 1-3 carry
// one axis each for both knots through LDS (pcr_axis_step), one step behind.  The adjoint solve (with_matrix = false) runs the
// right-hand sides alone with the saved multipliers, also through LDS (lane shuffles would have to cross the two knot halves).
// Before: 65+ pieces fell back to the one-lane-per-knot reduction with every wave idle but one per step.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pcr_waves2_wg(double *rowbuf, int nrow, int t, int N, bool with_matrix, double *pw, int pws, double *save, size_t sstride,
                                              size_t gk0, int nsteps, double *KV, double *KA) {
    double *rsb = rowbuf + (size_t)24 * nrow;
    const int wave = t >> 6, lane = t & 63, ax = wave - 1;
    int nst = 0;
    for (int s = 1; s < N - 1; s <<= 1) nst++;
    double D[2][4], L[2][4], U[2][4], r0[2] = {0.0, 0.0}, r1[2] = {0.0, 0.0};
    bool act[2]; int kc[2];
#pragma unroll
    for (int h = 0; h < 2; h++) { const int kk = lane + 64 * h; act[h] = kk >= 1 && kk <= N - 1; kc[h] = act[h] ? kk : 1; }
    if (wave == 0 && with_matrix) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const double2 l0 = MR2(0, 2, kc[h]), l1 = MR2(0, 3, kc[h]), u0 = MR2(0, 4, kc[h]), u1 = MR2(0, 5, kc[h]), d0 = MR2(0, 0, kc[h]), d1 = MR2(0, 1, kc[h]);
            L[h][0] = l0.x; L[h][1] = l0.y; L[h][2] = l1.x; L[h][3] = l1.y; U[h][0] = u0.x; U[h][1] = u0.y; U[h][2] = u1.x; U[h][3] = u1.y;
            D[h][0] = d0.x; D[h][1] = d0.y; D[h][2] = d1.x; D[h][3] = d1.y;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double I[4];
            m2_inv(D[h], I);
            if (act[h]) { MR2(0, 0, lane + 64 * h) = make_double2(I[0], I[1]); MR2(0, 1, lane + 64 * h) = make_double2(I[2], I[3]); }
        }
    } else if (wave >= 1) {
#pragma unroll
        for (int h = 0; h < 2; h++) { r0[h] = RS(0, ax, kc[h]); r1[h] = RS(0, 3 + ax, kc[h]); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int lag = with_matrix ? 1 : 0;                            // the right-hand side runs one step behind the matrix
    for (int it = 0; it < nst + lag; it++) {
        if (wave == 0) {
            if (with_matrix && it < nst) {
                const int s = 1 << it, buf = it & 1;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int kk = lane + 64 * h;
                    const bool inlo = act[h] && kk - s >= 1, inhi = act[h] && kk + s <= N - 1;
                    const int klo = inlo ? kk - s : kc[h], khi = inhi ? kk + s : kc[h];
                    double2 q[12];
#pragma unroll
                    for (int f = 0; f < 6; f++) { q[f] = MR2(buf, f, klo); q[6 + f] = MR2(buf, f, khi); }
                    double iLo[4], lL[4], lU[4], iHi[4], hL[4], hU[4], al[4], be[4], tt[4];
                    iLo[0] = inlo ? q[0].x : 1.0; iLo[1] = inlo ? q[0].y : 0.0; iLo[2] = inlo ? q[1].x : 0.0; iLo[3] = inlo ? q[1].y : 1.0;
                    lL[0] = inlo ? q[2].x : 0.0; lL[1] = inlo ? q[2].y : 0.0; lL[2] = inlo ? q[3].x : 0.0; lL[3] = inlo ? q[3].y : 0.0;
                    lU[0] = inlo ? q[4].x : 0.0; lU[1] = inlo ? q[4].y : 0.0; lU[2] = inlo ? q[5].x : 0.0; lU[3] = inlo ? q[5].y : 0.0;
                    iHi[0] = inhi ? q[6].x : 1.0; iHi[1] = inhi ? q[6].y : 0.0; iHi[2] = inhi ? q[7].x : 0.0; iHi[3] = inhi ? q[7].y : 1.0;
                    hL[0] = inhi ? q[8].x : 0.0; hL[1] = inhi ? q[8].y : 0.0; hL[2] = inhi ? q[9].x : 0.0; hL[3] = inhi ? q[9].y : 0.0;
                    hU[0] = inhi ? q[10].x : 0.0; hU[1] = inhi ? q[10].y : 0.0; hU[2] = inhi ? q[11].x : 0.0; hU[3] = inhi ? q[11].y : 0.0;
                    m2_mul(L[h], iLo, al);                              // pcr_step_inv, matrix part
                    m2_mul(U[h], iHi, be);
                    m2_mul(al, lU, tt);
#pragma unroll
                    for (int i = 0; i < 4; i++) D[h][i] = D[h][i] - tt[i];
                    m2_mul(be, hL, tt);
#pragma unroll
                    for (int i = 0; i < 4; i++) D[h][i] -= tt[i];
                    m2_mul(al, lL, tt);
#pragma unroll
                    for (int i = 0; i < 4; i++) L[h][i] = -tt[i];
                    m2_mul(be, hU, tt);
#pragma unroll
                    for (int i = 0; i < 4; i++) U[h][i] = -tt[i];
                    double I[4];
                    m2_inv(D[h], I);
                    if (act[h]) {
                        MR2(buf ^ 1, 0, kk) = make_double2(I[0], I[1]); MR2(buf ^ 1, 1, kk) = make_double2(I[2], I[3]);
                        MR2(buf ^ 1, 2, kk) = make_double2(L[h][0], L[h][1]); MR2(buf ^ 1, 3, kk) = make_double2(L[h][2], L[h][3]);
                        MR2(buf ^ 1, 4, kk) = make_double2(U[h][0], U[h][1]); MR2(buf ^ 1, 5, kk) = make_double2(U[h][2], U[h][3]);
#pragma unroll
                        for (int i = 0; i < 4; i++) { pw[kk * pws + it * 8 + i] = al[i]; pw[kk * pws + it * 8 + 4 + i] = be[i]; }
                        double2 *sp = save ? (double2 *)(save + (gk0 + kk) * sstride) : nullptr;
                        if (sp) {
                            sp[it * 4 + 0] = make_double2(al[0], al[1]); sp[it * 4 + 1] = make_double2(al[2], al[3]);
                            sp[it * 4 + 2] = make_double2(be[0], be[1]); sp[it * 4 + 3] = make_double2(be[2], be[3]);
                        }
                        if (it == nst - 1) {
#pragma unroll
                            for (int i = 0; i < 4; i++) pw[kk * pws + nsteps * 8 + i] = I[i];
                            if (sp) { sp[nsteps * 4] = make_double2(I[0], I[1]); sp[nsteps * 4 + 1] = make_double2(I[2], I[3]); }
                        }
                    }
                }
            }
        } else {
            const int st = it - lag;
            if (st >= 0 && st < nst) {
#pragma unroll
                for (int h = 0; h < 2; h++) pcr_axis_step(rsb, nrow, lane + 64 * h, ax, N, 1 << st, st, pw, pws, r0[h], r1[h]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (wave >= 1) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int kk = lane + 64 * h;
            if (act[h]) {
                const double *Di = pw + kk * pws + nsteps * 8;
                KV[ax * (nrow + 1) + kk] = Di[0] * r0[h] + Di[1] * r1[h];
                KA[ax * (nrow + 1) + kk] = Di[2] * r0[h] + Di[3] * r1[h];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Forward reduction for <= 64 pieces, round-2 form: the matrix wave runs FREE.  All interior knots live in the 64 lanes of wave 0,
// so its six reduction steps need no workgroup barrier at all (LDS operations of one wave complete in order); it publishes a
// step counter in LDS behind each step's multipliers, and the three axis waves - which meanwhile evaluate the waypoint map and
// build their right-hand sides - follow it step by step, exchanging neighbours by lane shifts.  Before, the matrix steps started
// only after the waypoint map and the row assembly of the whole workgroup and every step ended in a workgroup barrier:
// 11.1 k of the forward map's 19.4 k cycles.  Per step the wave also (i) takes the rows beyond the ends from an IDENTITY row (row 0,
// knot 0 is not an unknown) instead of selecting 24 doubles lane by lane, (ii) uses the symmetry of the system, U_{k-s} = L_k^T and
// L_{k+s} = U_k^T at every level (frx_minco.hpp: K is the Hessian of the jerk energy), instead of reading those blocks, and
// (iii) inverts the 2x2 diagonal block with rcp_fast.  (The symmetry goes all the way: U is not even part of a row's state.)
// ---------------------------------------------------------------------------------------------
typedef volatile __attribute__((address_space(3))) unsigned *lds_vuptr;
// (bounded like every other spin of the library: the matrix wave waits for nobody but the one barrier the axis waves reach without waiting, so the
// bound - about a second - can only expire on a fault; the evaluation then finishes with whatever the multipliers hold instead of hanging the GPU)
__device__ __forceinline__ void lds_wait_ge(lds_vuptr w, unsigned want) {
    for (unsigned spins = 0; *w < want && spins < (1u << 24); spins++) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// wave 0 of the workgroup; kk = lane = knot.  hL / hR: durations of the pieces left and right of the knot (1.0 on lanes without a knot).
// Executes exactly ONE s_barrier (after its second step - or its last, or the final inverse, when there are fewer): the caller's axis
// waves meet it there once their knot positions are in LDS (measured: they arrive ~2.6 k cycles after the durations are known, the
// matrix wave's build + first step take 1.5 k).
__device__ __forceinline__ void pcr_matrix_wave64(double *rowbuf, int kk, int N, double hL, double hR, double *pw, int pws, double *save, size_t sstride,
                                                  size_t gk0, int nsteps, lds_vuptr progress) {
    const int nrow = 64;
    const bool act = kk >= 1 && kk <= N - 1;
    int nst = 0;
    for (int s = 1; s < N - 1; s <<= 1) nst++;
    // State of a row: D and L only.  U is never kept: at every level U_k = L_{k+s}^T (the system is symmetric), so the block a step needs
    // is the transpose of the L it reads from the row above anyway, and the U recurrence (a 2x2 product, two LDS writes) disappears.
    double D[4], L[4], I[4];
    {
        KnotRow me;
        knot_row_matrix(hL, hR, me);
#pragma unroll
        for (int i = 0; i < 4; i++) { D[i] = me.D[i]; L[i] = kk == 1 ? 0.0 : me.L[i]; }   // fixed head knot: its coupling is on the right-hand side (the tail's too: no row N)
    }
    m2_inv_fast(D, I);
    if (kk == 0) {                                                  // the identity row (D^-1 = 1, L = 0) stands in for the rows beyond both ends, in both buffers
#pragma unroll
        for (int buf = 0; buf < 2; buf++) {
            MR2(buf, 0, 0) = make_double2(1.0, 0.0); MR2(buf, 1, 0) = make_double2(0.0, 1.0);
            MR2(buf, 2, 0) = make_double2(0.0, 0.0); MR2(buf, 3, 0) = make_double2(0.0, 0.0);
        }
    } else if (act) {
        MR2(0, 0, kk) = make_double2(I[0], I[1]); MR2(0, 1, kk) = make_double2(I[2], I[3]);
        MR2(0, 2, kk) = make_double2(L[0], L[1]); MR2(0, 3, kk) = make_double2(L[2], L[3]);
    }
    double2 *sp = save ? (double2 *)(save + (gk0 + kk) * sstride) : nullptr;     // null: the multipliers stay in LDS (resident caller)
    // (Round 5 tried the neighbours' (D^-1, L) straight out of their lanes' registers through the LDS crossbar - 32 ds_bpermute per step instead of four
    // 16-byte LDS writes, a wait and eight 16-byte reads: bit-identical and SLOWER, the matrix wave done after 10.1 k cycles instead of 9.2 k, stage
    // kernel 8.05 against 7.60 us.  profiles/NOTES.md)
    for (int it = 0; it < nst; it++) {
        const int s = 1 << it, buf = it & 1;
        const bool inlo = act && kk - s >= 1, inhi = act && kk + s <= N - 1;
        const int klo = inlo ? kk - s : 0, khi = inhi ? kk + s : 0;
        const double2 a0 = MR2(buf, 0, klo), a1 = MR2(buf, 1, klo), a2 = MR2(buf, 2, klo), a3 = MR2(buf, 3, klo);
        const double2 b0 = MR2(buf, 0, khi), b1 = MR2(buf, 1, khi), b2 = MR2(buf, 2, khi), b3 = MR2(buf, 3, khi);
        const double iLo[4] = {a0.x, a0.y, a1.x, a1.y}, lL[4] = {a2.x, a2.y, a3.x, a3.y};
        const double iHi[4] = {b0.x, b0.y, b1.x, b1.y}, hL[4] = {b2.x, b2.y, b3.x, b3.y};
        const double U[4] = {hL[0], hL[2], hL[1], hL[3]};           // this row's U = (L of the row above)^T
        const double Lt[4] = {L[0], L[2], L[1], L[3]};              // = U of the row below
        double al[4], be[4], tt[4];
        m2_mul(L, iLo, al);                                         // pcr_step_inv, matrix part
        m2_mul(U, iHi, be);
        m2_mul(al, Lt, tt);
#pragma unroll
        for (int i = 0; i < 4; i++) D[i] = D[i] - tt[i];
        m2_mul(be, hL, tt);
#pragma unroll
        for (int i = 0; i < 4; i++) D[i] -= tt[i];
        m2_mul(al, lL, tt);
#pragma unroll
        for (int i = 0; i < 4; i++) L[i] = -tt[i];
        m2_inv_fast(D, I);
        if (act) {
            MR2(buf ^ 1, 0, kk) = make_double2(I[0], I[1]); MR2(buf ^ 1, 1, kk) = make_double2(I[2], I[3]);
            MR2(buf ^ 1, 2, kk) = make_double2(L[0], L[1]); MR2(buf ^ 1, 3, kk) = make_double2(L[2], L[3]);
#pragma unroll
            for (int i = 0; i < 4; i++) { pw[kk * pws + it * 8 + i] = al[i]; pw[kk * pws + it * 8 + 4 + i] = be[i]; }
            if (sp) {
                sp[it * 4 + 0] = make_double2(al[0], al[1]); sp[it * 4 + 1] = make_double2(al[2], al[3]);
                sp[it * 4 + 2] = make_double2(be[0], be[1]); sp[it * 4 + 3] = make_double2(be[2], be[3]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kk == 0) *progress = (unsigned)(it + 1);
        if (it == (nst > 1 ? 1 : 0)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the axis waves need about two steps' time for the waypoint map
    }
    if (act) {                                                      // D^-1 of the decoupled rows
#pragma unroll
        for (int i = 0; i < 4; i++) pw[kk * pws + nsteps * 8 + i] = I[i];
        if (sp) { sp[nsteps * 4] = make_double2(I[0], I[1]); sp[nsteps * 4 + 1] = make_double2(I[2], I[3]); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (kk == 0) *progress = (unsigned)(nst + 1);
    if (nst == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#undef MR2
#undef RS

// SH: T and C are consumed by OTHER workgroups of the same launch (see ldg / stg above).
// NR > 0: the caller KNOWS the geometry class at compile time (nrow = NR rows, 256 threads: the resident round kernel's instantiation for <= 64 pieces):
// the forms for the other classes are not compiled into it - k_round carried all three (146 KB of code against a 64 KB instruction cache that two
// CUs share; the leader walks through forward map, adjoint and control code once per round, each time from L2).
// MODE (bits; 0 for the stage kernels): 1 = the caller hands in resident operands (`ro`, not null) that THIS call fills - x and the polytopes are staged into
// ro->xs / ro->vs here, with the index-table loads of the body in the same memory latency (the leader of the one-launch evaluation, frx_eval_kernel.hpp, keeps
// them for the adjoint of the same launch); 2 = with `go` not null (GranuleOut) the coefficients and durations leave as granules instead of plain stores (the
// one-launch evaluation; measured inside the resident round kernel as well - slower there than its drained phase word, profiles/NOTES.md); 4 = no global (C, T) at all: they
// stay in ct_lds (the solo launch, whose penalty phase and adjoint read that copy).  What a caller does not ask for is not compiled into it.
// RB > 0 (with NR = 64 and a ct_lds copy only): the caller's scratch holds RB doubles of row buffer instead of 36 nrow - the wave-specialised path uses the (D^-1, L)
// rows of two buffers ([0, 1280) doubles) and the progress words at 24 nrow = 1536; the coefficients are collected in ct_lds (the solo launch, frx_solo_kernel.hpp: LDS is
// what keeps a second workgroup off its CU).
template <bool SH, int NR = 0, int MODE = 0, int RB = 0>
__device__ __forceinline__ void forward_knot_body(const DevProblem &dp, const double *__restrict__ x, double *__restrict__ Tout, double *__restrict__ Cout,
                               int maxCN, int maxXb, int maxVb, int nrow_rt, double *__restrict__ pcrw, int nsteps, int b, double *sm, double *ct_lds = nullptr, bool wt = true, const ResidentOps *ro = nullptr,
                               const KnotPre *pre = nullptr, const GranuleOut *go = nullptr) {
    // ct_lds (optional, LDS, 19 doubles per piece: 18 coefficients + duration): a copy for the backward pass of the same workgroup
    const int nrow = NR > 0 ? NR : nrow_rt;
    const int k = threadIdx.x, nthr = NR > 0 ? 256 : (int)blockDim.x;
    const int p0 = pre ? pre->p0 : dp.poff[b], N = pre ? pre->N : dp.poff[b + 1] - p0;
    const int c0 = pre ? pre->c0 : dp.coff[b], cN = pre ? pre->cN : dp.coff[b + 1] - c0;
    const int x0 = pre ? pre->x0 : dp.xoff[b];
    double *rowbuf = sm;
    static_assert(RB == 0 || (NR == 64 && RB >= 24 * 64 + 2), "a short row buffer: the <= 64-piece path only, progress words included");
    double *KP = rowbuf + (RB > 0 ? (size_t)RB : (size_t)36 * nrow);          // knot positions  [3][nthr+1]
    double *KV = KP + 3 * (nrow + 1);
    double *KA = KV + 3 * (nrow + 1);
    double *Tf = KA + 3 * (nrow + 1);
    double *Tc = Tf + nrow;
    double *xs = Tc + maxCN;                          // this candidate's variables (tau, xi), staged once
    double *vs = xs + maxXb;                          // this candidate's waypoint polytopes [v0, edges], waypoint order
    double *pwf = vs + maxVb;                          // [nrow][nsteps*8+5] reduction multipliers (wave-specialised path)
    if (ro) { xs = ro->xs; vs = ro->vs; if (nrow == 64 && nthr == 256) pwf = ro->pw; }
#define KN(arr, axis, idx) arr[(axis) * (nrow + 1) + (idx)]
#define FWD_STAMP(slot) do { if (!(ro && ro->quiet)) FRX_STAMP(slot); } while (0)
#define FWD_STAMP_AX(slot) do { if (!(ro && ro->quiet)) FRX_STAMP_AX(slot); } while (0)
    FWD_STAMP(0);
    // Every global read of the kernel is issued here, before the first barrier, so the whole kernel pays ONE memory
    // latency (loads placed in later phases cannot be hoisted over the barriers by the compiler: measured +4 us).
    int r_pc = 0, r_piv = 1, r_wnv = 1, r_wvb = 0, r_wxb = 0;
    // (first vertex of this candidate: the compiler fetches dp.* through VECTOR loads - it cannot prove that nothing in the kernel writes
    // there - and where the expression stood in the waypoint loops below, each axis wave paid a trip to L2 in the middle of its chain)
    const int cv0 = pre ? pre->cv0 : __builtin_amdgcn_readfirstlane(dp.cvoff[b]);
    double r_bs[6] = {0, 0, 0, 0, 0, 0};
    if (pre) { r_pc = pre->pc; r_piv = pre->piv; }
    else if (k < N) { r_pc = dp.piece_coarse[p0 + k]; r_piv = dp.piece_iv[p0 + k]; }
    // waypoint w is handled by the quad k >> 2 - or, when wave 0 is the free-running matrix wave (wsp64 below), by the PAIR (k - 64) >> 1
    const bool wsp64 = nrow == 64 && nthr == 256;
    const int t2 = k - 64;                                         // index among the axis waves
    const int wq = wsp64 ? (t2 >= 0 ? (t2 >> 1) : N) : (k >> 2);
    if (pre) { r_wnv = pre->wnv; r_wvb = pre->wvb; r_wxb = pre->wxb; }
    else if (wq < N - 1) { const int gw = p0 - b + wq; r_wnv = dp.wp_nv[gw]; r_wvb = dp.wp_vbeg[gw]; r_wxb = dp.wp_xbeg[gw]; }
    const int kbs = wsp64 ? t2 : k;                                // the three threads that place the fixed end states
    if (pre) { r_bs[0] = pre->bs[0]; r_bs[1] = pre->bs[1]; r_bs[2] = pre->bs[2]; r_bs[3] = pre->bs[3]; r_bs[4] = pre->bs[4]; r_bs[5] = pre->bs[5]; }
    else if (kbs >= 0 && kbs < 3) {
#pragma unroll
        for (int q = 0; q < 3; q++) { r_bs[q] = dp.headPVA[b * 9 + 3 * q + kbs]; r_bs[3 + q] = dp.tailPVA[b * 9 + 3 * q + kbs]; }
    }
    lds_vuptr progress = (lds_vuptr)(unsigned *)(rowbuf + (size_t)24 * nrow);   // step counter of the matrix wave (wsp64; the rhs buffers behind the rows are unused there)
    if (wsp64 && k == 0) *progress = 0u;
    {   // coalesced staging: every later access is an LDS access (the per-waypoint loops would otherwise serialise
        // one global-memory latency per vertex)
        if (!ro || (MODE & 1)) {
            const int nx = dp.xoff[b + 1] - x0;
            const int v0 = dp.cvoff[b], nvd = 3 * (dp.cvoff[b + 1] - v0);
            const double *vsrc = dp.vrec + 3 * (size_t)v0;
            stage_to_lds<4>(xs, x + x0, nx, k, nthr);
            stage_to_lds<8>(vs, vsrc, nvd, k, nthr);
        }
    }
    // (Round 5: with a resident caller, <= 64 pieces, soft total time the two workgroup barriers in front of the matrix wave order nothing - nothing is staged,
    // the durations are produced and consumed by wave 0 alone - and a build without them is bit-identical; it is also no faster: the duration is ready after
    // 1.3 k cycles either way.  What that stretch WAS waiting for were the index-table loads, see KnotPre.)
    __syncthreads();
    FWD_STAMP(1);

    // forwardT (CPU.hpp:626-676)
    if (dp.soft) {
        for (int i = k; i < cN; i += nthr) Tc[i] = tau_to_T(xs[i], dp.c2 != 0);
    } else if (k == 0) {
        const int Ms1 = cN - 1;
        double sum = 0.0;
        for (int i = 0; i < Ms1; i++) Tc[i] = tau_to_T(xs[i], dp.c2 != 0);
        Tc[Ms1] = 0.0;
        for (int i = 0; i <= Ms1; i++) sum += Tc[i];
        const double den = 1.0 + sum;
        for (int i = 0; i <= Ms1; i++) Tc[i] /= den;
        sum = 0.0;
        for (int i = 0; i <= Ms1; i++) sum += Tc[i];
        Tc[Ms1] = 1.0 - sum;
        for (int i = 0; i <= Ms1; i++) Tc[i] *= dp.sumT;
    }
    __syncthreads();
    // splitToFineT (CPU.hpp:930-944)
    double hMine = 1.0;
    if (k < N) {
        hMine = Tc[r_pc - c0] / r_piv;
        Tf[k] = hMine;
        if ((MODE & 2) && SH && go && go->ll) rk_ll_put(go->ll + 2 * ((size_t)(p0 + k) * 19 + 18), hMine, go->tag, go->mxw ? true : wt);   // (mxw: where the consumers run is not known yet - early, off the critical path: write-through)
        else if (!(MODE & 4)) stg<SH>(Tout + p0 + k, hMine, wt);             // (MODE & 4: (C, T) stay in the caller's ct_lds copy and never go to global memory - the solo launch)
        if (ct_lds) ct_lds[k * 19 + 18] = hMine;
    }
    FWD_STAMP(2);
    if (wsp64) {
        // ---- <= 64 pieces: wave 0 = matrix wave (pcr_matrix_wave64), waves 1-3 = one axis each ----
        const int wave = __builtin_amdgcn_readfirstlane(k >> 6), kk = k & 63;
        const int pws = nsteps * 8 + 5;
        double *cstage = ct_lds ? ct_lds : rowbuf;                         // where the coefficients are collected (the matrix rows are dead by the time the axis waves write here)
        if (wave == 0) {
            // durations left and right of knot kk: the left one comes from the neighbouring lane (lane = piece = knot)
            const double hLs = lane_up1(hMine);
            const bool act0 = kk >= 1 && kk <= N - 1;
            pcr_matrix_wave64(rowbuf, kk, N, act0 ? hLs : 1.0, act0 ? hMine : 1.0, pwf, pws, ro ? nullptr : pcrw, (size_t)(nsteps * 8 + 4), (size_t)p0, nsteps, progress);
            FWD_STAMP(5);
        } else {
            const int ax = wave - 1;
            // forwardP (CPU.hpp:729-747) on PAIRS of lanes of the axis waves: q = v0 + (2/(1+|xi|^2))^2 * sum_a V_a xi_a^2
            for (int w0 = 0; w0 < N - 1; w0 += 96) {
                const int w = w0 + (t2 >> 1), sub = t2 & 1;
                double nrm = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0;
                const bool wact = w < N - 1;
                const double *V = vs, *xi = xs;
                if (wact) {
                    int wnv = r_wnv, wvb = r_wvb, wxb = r_wxb;          // prefetched for the first (usually only) pass
                    if (w0 > 0) { const int gw = p0 - b + w; wnv = dp.wp_nv[gw]; wvb = dp.wp_vbeg[gw]; wxb = dp.wp_xbeg[gw]; }
                    const int nv1 = wnv - 1;
                    V = vs + 3 * (wvb - cv0) + (ro ? ro->vskew * w : 0);
                    xi = xs + (wxb - x0);
                    for (int a0 = sub; a0 < nv1; a0 += 8) {               // four vertices per trip, their LDS reads in flight together (clamped, not predicated)
                        double xv[4], v0[4], v1[4], v2[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) { const int a = min(a0 + 2 * j, nv1 - 1); xv[j] = xi[a]; v0[j] = V[3 * (a + 1)]; v1[j] = V[3 * (a + 1) + 1]; v2[j] = V[3 * (a + 1) + 2]; }
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (a0 + 2 * j < nv1) { const double x2 = xv[j] * xv[j]; nrm += x2; q0 += v0[j] * x2; q1 += v1[j] * x2; q2 += v2[j] * x2; }
                    }
                }
                nrm += dpp_mov<0xB1>(nrm); q0 += dpp_mov<0xB1>(q0); q1 += dpp_mov<0xB1>(q1); q2 += dpp_mov<0xB1>(q2);   // pair sums
                if (wact && sub == 0) {
                    if (ro && ro->wq) { double *wq = ro->wq + 4 * w; wq[0] = nrm; wq[1] = q0; wq[2] = q1; wq[3] = q2; }
                    else if (dp.wq_glob) { double2 *wq = (double2 *)(dp.wq_glob + 4 * (size_t)(p0 - b + w)); wq[0] = make_double2(nrm, q0); wq[1] = make_double2(q1, q2); }
                    const double sc = 2.0 / (1.0 + nrm), sc2 = sc * sc;
                    KN(KP, 0, w + 1) = sc2 * q0 + V[0]; KN(KP, 1, w + 1) = sc2 * q1 + V[1]; KN(KP, 2, w + 1) = sc2 * q2 + V[2];
                }
            }
            if (t2 < 3) {                                  // fixed head / tail knot states (CPU.hpp:440-442, 497-499)
                KN(KP, t2, 0) = r_bs[0]; KN(KV, t2, 0) = r_bs[1]; KN(KA, t2, 0) = r_bs[2];
                KN(KP, t2, N) = r_bs[3]; KN(KV, t2, N) = r_bs[4]; KN(KA, t2, N) = r_bs[5];
            }
            FWD_STAMP_AX(8);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // meets the matrix wave's only barrier: knot positions and Tf are in LDS
            FWD_STAMP_AX(9);
            // right-hand side of knot kk for this axis (knot_row_rhs; the fixed end states move to the right-hand side)
            const bool act = kk >= 1 && kk <= N - 1;
            const int kc = act ? kk : 1;
            double r0 = 0.0, r1 = 0.0;
            if (act) {
                const double hL = Tf[kk - 1], hR = Tf[kk];
                knot_row_rhs(hL, hR, KN(KP, ax, kk) - KN(KP, ax, kk - 1), KN(KP, ax, kk + 1) - KN(KP, ax, kk), r0, r1);
                if (kk == 1 || kk == N - 1) {
                    KnotRow m0;
                    knot_row_matrix(hL, hR, m0);
                    if (kk == 1) {
                        r0 -= m0.L[0] * KN(KV, ax, 0) + m0.L[1] * KN(KA, ax, 0);
                        r1 -= m0.L[2] * KN(KV, ax, 0) + m0.L[3] * KN(KA, ax, 0);
                    }
                    if (kk == N - 1) {
                        r0 -= m0.U[0] * KN(KV, ax, N) + m0.U[1] * KN(KA, ax, N);
                        r1 -= m0.U[2] * KN(KV, ax, N) + m0.U[3] * KN(KA, ax, N);
                    }
                }
            }
            FWD_STAMP_AX(10);
            if ((MODE & 2) && SH && go && go->ll && go->mxw && wave == 3) {
                // Do all consumers of the granules run on this XCD?  They said so, or not yet, in go->mxw[0 .. nmx) (write-through stores at their entry, microseconds ago);
                // this wave has the slack for the trip - its reduction below follows the matrix wave, which is two steps into its six by now.  "Not yet" counts as no.
                unsigned vq = go->gate_val;
                if (kk < go->nmx) vq = __hip_atomic_load(go->mxw + kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool all_here = __builtin_amdgcn_ballot_w64(vq != go->gate_val) == 0ull;
                if (kk == 0) progress[1] = all_here ? 1u : 0u;
            }
            int nst = 0;
            for (int s = 1; s < N - 1; s <<= 1) nst++;
            for (int st = 0; st < nst; st++) {                             // pcr_rhs_step behind the matrix wave, neighbours by lane shifts
                lds_wait_ge(progress, (unsigned)(st + 1));
                const int s = 1 << st;
                const bool inlo = act && kk - s >= 1, inhi = act && kk + s <= N - 1;
                double ab[8];
#pragma unroll
                for (int i = 0; i < 8; i++) ab[i] = pwf[kc * pws + st * 8 + i];
                const double l0r = __shfl_up(r0, s, 64), l1r = __shfl_up(r1, s, 64), h0r = __shfl_down(r0, s, 64), h1r = __shfl_down(r1, s, 64);
                const double l0 = inlo ? l0r : 0.0, l1 = inlo ? l1r : 0.0, h0 = inhi ? h0r : 0.0, h1 = inhi ? h1r : 0.0;
                const double n0 = r0 - (ab[0] * l0 + ab[1] * l1) - (ab[4] * h0 + ab[5] * h1);
                const double n1 = r1 - (ab[2] * l0 + ab[3] * l1) - (ab[6] * h0 + ab[7] * h1);
                r0 = n0; r1 = n1;
            }
            FWD_STAMP_AX(11);
            // everything of the Hermite stage that does not need the solution is computed while the matrix wave finishes: durations and their
            // powers, the position part of the coefficients (hermite_coeffs with v = a = 0), the fixed end states
            const int kp = kk < N ? kk : 0;
            const double hK = Tf[kp], pK = KN(KP, ax, kp), pR = KN(KP, ax, kp + 1);
            const double ih = rcp_fast(hK), ih2 = ih * ih, ih3 = ih2 * ih, ih4 = ih2 * ih2, ih5 = ih4 * ih, dl = pR - pK;
            const double c3p = 10.0 * dl * ih3, c4p = -15.0 * dl * ih4, c5p = 6.0 * dl * ih5;
            const double vHead = KN(KV, ax, 0), aHead = KN(KA, ax, 0), vTail = KN(KV, ax, N), aTail = KN(KA, ax, N);
            lds_wait_ge(progress, (unsigned)(nst + 1));
            double vK, aK;                                                 // (v, a) of knot kk on this axis
            {
                const double *Di = pwf + kc * pws + nsteps * 8;
                vK = Di[0] * r0 + Di[1] * r1; aK = Di[2] * r0 + Di[3] * r1;
            }
            if (kk == 0) { vK = vHead; aK = aHead; }
            double vR = lane_down1(vK), aR = lane_down1(aK);
            if (kk == N - 1) { vR = vTail; aR = aTail; }
            // piece coefficients of this axis (quintic Hermite, hermite_coeffs term by term): piece kk between knots kk and kk + 1
            if (kk < N) {
                double cq[6];
                cq[0] = pK; cq[1] = vK; cq[2] = 0.5 * aK;
                cq[3] = c3p - (4.0 * vR + 6.0 * vK) * ih2 - 0.5 * (3.0 * aK - aR) * ih;
                cq[4] = c4p + (7.0 * vR + 8.0 * vK) * ih3 + 0.5 * (3.0 * aK - 2.0 * aR) * ih2;
                cq[5] = c5p - 3.0 * (vR + vK) * ih4 - 0.5 * (aK - aR) * ih3;
                // to LDS only: 18 doubles per piece (stride 19), the caller's (C, T) copy when it keeps one, the dead row buffer otherwise
#pragma unroll
                for (int q = 0; q < 6; q++) cstage[kk * 19 + q * 3 + ax] = cq[q];
            }
            FWD_STAMP_AX(12);
        }
        // C leaves the workgroup as ONE coalesced sweep of 16-byte stores by all four waves.  Stored straight from the axis lanes it was 1152
        // scattered 8-byte write-through stores (lane stride 144 bytes), and draining them cost the resident kernel ~2 us per round.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const bool wt_ll = ((MODE & 2) && SH && go && go->ll && go->mxw) ? progress[1] == 0u : wt;   // (granules: see the decision of wave 3 above)
        if (!(MODE & 4))
        for (int i = k; i < 9 * N; i += 256) {                              // 9 pairs of doubles per piece
            const int pc = i / 9, q2 = 2 * (i - 9 * pc);
            const double v0 = cstage[pc * 19 + q2], v1 = cstage[pc * 19 + q2 + 1];
            double *dst = Cout + (size_t)(p0 + pc) * 18 + q2;
            if ((MODE & 2) && SH && go && go->ll) { ll_u64 *gl = go->ll + 2 * ((size_t)(p0 + pc) * 19 + q2); rk_ll_put(gl, v0, go->tag, wt_ll); rk_ll_put(gl + 2, v1, go->tag, wt_ll); }
            else if (SH && wt) { stg<SH>(dst, v0, true); stg<SH>(dst + 1, v1, true); }
            else *(double2 *)dst = make_double2(v0, v1);
        }
        // the gate of the granules' consumers: set BEHIND the sweep, not drained - it tells the members' waves that a poll of their granules is now worth its trip
        // (polled from the start of the launch, by 49 k lanes, the granules cost more than they save: every poll is a transaction on the fabric between the XCDs)
        if ((MODE & 2) && SH && go && go->ll && go->gate && k == 255) __hip_atomic_store(go->gate, go->gate_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        FWD_STAMP(6);
        return;
    }
    // forwardP (CPU.hpp:729-747): waypoint w (= knot w+1) is handled by a QUAD of lanes, each taking every 4th vertex;
    // q = v0 + (2/(1+|xi|^2))^2 * sum_a V_a xi_a^2 — one pass over the vertices, quad sums by DPP-class shuffles
    for (int w0 = 0; w0 < N - 1; w0 += nthr / 4) {
        const int w = w0 + (k >> 2), sub = k & 3;
        double nrm = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0;
        const bool wact = w < N - 1;
        const double *V = vs, *xi = xs;
        int nv1 = 0;
        if (wact) {
            int wnv = r_wnv, wvb = r_wvb, wxb = r_wxb;                  // prefetched for the first (usually only) pass
            if (w0 > 0) { const int gw = p0 - b + w; wnv = dp.wp_nv[gw]; wvb = dp.wp_vbeg[gw]; wxb = dp.wp_xbeg[gw]; }
            nv1 = wnv - 1;
            V = vs + 3 * (wvb - cv0) + (ro ? ro->vskew * w : 0);
            xi = xs + (wxb - x0);
            for (int a = sub; a < nv1; a += 4) {
                const double x2 = xi[a] * xi[a];
                nrm += x2;
                q0 += V[3 * (a + 1)] * x2; q1 += V[3 * (a + 1) + 1] * x2; q2 += V[3 * (a + 1) + 2] * x2;
            }
        }
        nrm = quad_sum(nrm); q0 = quad_sum(q0); q1 = quad_sum(q1); q2 = quad_sum(q2);
        if (wact && sub == 0) {
            const double sc = 2.0 / (1.0 + nrm), sc2 = sc * sc;
            KN(KP, 0, w + 1) = sc2 * q0 + V[0]; KN(KP, 1, w + 1) = sc2 * q1 + V[1]; KN(KP, 2, w + 1) = sc2 * q2 + V[2];
        }
    }
    if (k < 3) {                                       // fixed head / tail knot states (CPU.hpp:440-442, 497-499)
        KN(KP, k, 0) = r_bs[0]; KN(KV, k, 0) = r_bs[1]; KN(KA, k, 0) = r_bs[2];
        KN(KP, k, N) = r_bs[3]; KN(KV, k, N) = r_bs[4]; KN(KA, k, N) = r_bs[5];
    }
    __syncthreads();

    FWD_STAMP(3);
    // knot system rows + PCR
    KnotRow me;
    double vk[3] = {0, 0, 0}, ak[3] = {0, 0, 0};
    if (k >= 1 && k <= N - 1) {
        const double hL = Tf[k - 1], hR = hMine;
        knot_row_matrix(hL, hR, me);
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
            knot_row_rhs(hL, hR, KN(KP, ax, k) - KN(KP, ax, k - 1), KN(KP, ax, k + 1) - KN(KP, ax, k), me.r[ax], me.r[3 + ax]);
        if (k == 1) {
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                me.r[ax] -= me.L[0] * KN(KV, ax, 0) + me.L[1] * KN(KA, ax, 0);
                me.r[3 + ax] -= me.L[2] * KN(KV, ax, 0) + me.L[3] * KN(KA, ax, 0);
            }
            me.L[0] = me.L[1] = me.L[2] = me.L[3] = 0.0;
        }
        if (k == N - 1) {
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                me.r[ax] -= me.U[0] * KN(KV, ax, N) + me.U[1] * KN(KA, ax, N);
                me.r[3 + ax] -= me.U[2] * KN(KV, ax, N) + me.U[3] * KN(KA, ax, N);
            }
            me.U[0] = me.U[1] = me.U[2] = me.U[3] = 0.0;
        }
    }
    FWD_STAMP(4);
    if (nrow == 128 && nthr == 256) {                               // wave-specialised reduction with two knots per lane (<= 64 pieces: wsp64 above)
        if (k >= 1 && k <= N - 1) {
            double2 *mr = (double2 *)rowbuf;
            mr[(0 * 6 + 0) * nrow + k] = make_double2(me.D[0], me.D[1]); mr[(0 * 6 + 1) * nrow + k] = make_double2(me.D[2], me.D[3]);
            mr[(0 * 6 + 2) * nrow + k] = make_double2(me.L[0], me.L[1]); mr[(0 * 6 + 3) * nrow + k] = make_double2(me.L[2], me.L[3]);
            mr[(0 * 6 + 4) * nrow + k] = make_double2(me.U[0], me.U[1]); mr[(0 * 6 + 5) * nrow + k] = make_double2(me.U[2], me.U[3]);
#pragma unroll
            for (int i = 0; i < 6; i++) rowbuf[(size_t)24 * nrow + i * nrow + k] = me.r[i];
        }
        __syncthreads();
        pcr_waves2_wg(rowbuf, nrow, k, N, true, pwf, nsteps * 8 + 5, pcrw, (size_t)(nsteps * 8 + 4), (size_t)p0, nsteps, KV, KA);
    } else {
        pcr_solve_wg(rowbuf, nrow, k, N, me, vk, ak, pcrw, (size_t)(nsteps * 8 + 4), (size_t)(p0 + k), nsteps);
        if (k >= 1 && k <= N - 1) {
#pragma unroll
            for (int ax = 0; ax < 3; ax++) { KN(KV, ax, k) = vk[ax]; KN(KA, ax, k) = ak[ax]; }
        }
    }
    FWD_STAMP(5);
    __syncthreads();
    // piece coefficients (quintic Hermite), 18 contiguous doubles per piece
    if (k < N) {
        double *co = Cout + (size_t)(p0 + k) * 18;
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
            double c[6];
            hermite_coeffs(hMine, KN(KP, ax, k), KN(KV, ax, k), KN(KA, ax, k), KN(KP, ax, k + 1), KN(KV, ax, k + 1), KN(KA, ax, k + 1), c);
#pragma unroll
            for (int q = 0; q < 6; q++) { stg<SH>(co + q * 3 + ax, c[q], wt); if (ct_lds) ct_lds[k * 19 + q * 3 + ax] = c[q]; }
        }
    }
    FWD_STAMP(6);
}
#undef FWD_STAMP
#undef FWD_STAMP_AX
FRX_KERNEL_LINKAGE __global__ __launch_bounds__(256) void k_forward_knot(DevProblem dp, const double *__restrict__ x, double *__restrict__ Tout, double *__restrict__ Cout,
                               int maxCN, int maxXb, int maxVb, int nrow, double *__restrict__ pcrw, int nsteps) {
    extern __shared__ double sm[];
    if (dp.cand_active && !(dp.cand_active[blockIdx.x] & DV_EVAL)) return;
    forward_knot_body<false>(dp, x, Tout, Cout, maxCN, maxXb, maxVb, nrow, pcrw, nsteps, blockIdx.x, sm);
}
// the same for batches whose candidates all have <= 64 pieces: only the wave-specialised form is compiled in (a third of the code)
FRX_KERNEL_LINKAGE __global__ __launch_bounds__(256) void k_forward_knot64(DevProblem dp, const double *__restrict__ x, double *__restrict__ Tout, double *__restrict__ Cout,
                               int maxCN, int maxXb, int maxVb, double *__restrict__ pcrw, int nsteps) {
    extern __shared__ double sm[];
    if (dp.cand_active && !(dp.cand_active[blockIdx.x] & DV_EVAL)) return;
    forward_knot_body<false, 64>(dp, x, Tout, Cout, maxCN, maxXb, maxVb, 64, pcrw, nsteps, blockIdx.x, sm);
}

// ---------------------------------------------------------------------------------------------
// Adjoint for <= 64 pieces on 256 threads, round-2 form (same LDS layout as backward_knot_body, which calls it): ONE WAVE PER AXIS.
// Lane kk of axis wave a carries axis a of piece kk and of knot kk: jerk-energy gradient, Hermite adjoint, the knot system's adjoint
// solve with the saved multipliers and the knot adjoint all stay in its registers, and everything that crosses a piece boundary is a
// lane shift (end-of-piece adjoints to the next knot, mu of the next knot, d f / d(p_{k+1} - p_k) of the previous piece).  Wave 0
// meanwhile adds up jerk energy and penalty partials per piece.  One barrier later wave 0 owns the time side (gradient w.r.t. the
// durations, merge to coarse pieces, cost, tau layer) while the axis waves run the waypoint layer on pairs of lanes; a second barrier
// collects the line-search sums.  Before: every phase on wave 0 for all three axes in turn, eight workgroup barriers and the
// knot arrays through LDS in between - 21 k cycles, of which 5.5 k + 6.7 k in the knot adjoint and the layers.
// ---------------------------------------------------------------------------------------------
// The same adjoint in the order of rounds 2-4, for the STAGE kernels (no poll for the penalty partials: everything is loaded up front and the scheduler
// orders one straight-line body itself).  The resident form below - the partial-independent work hoisted in front of the poll by hand - measured 0.4 us
// SLOWER as a stage kernel (8.00 against 7.62 us, build against build on one box, round 5), so each caller keeps the order that suits it; the arithmetic
// is expression for expression the same.
template <bool SH, int RB = 0>
__device__ __forceinline__ void backward_knot_wsp64_stage(const DevProblem &dp, const double *__restrict__ x, const double *__restrict__ Tin,
                                const double *__restrict__ Cin, const double *__restrict__ out20, double *__restrict__ f,
                                double *__restrict__ g, int maxCN, int maxXb, int maxVb, const double *__restrict__ pcrw, int nsteps,
                                const LineSearchTap &tap, int b, double *sm, const double *ct_lds, const ResidentOps *ro) {
    const int nrow = 64, nthr = 256;
    const int k = threadIdx.x, kk = k & 63, t2 = k - 64;
    const int wave = __builtin_amdgcn_readfirstlane(k >> 6);
    const int p0 = dp.poff[b], N = dp.poff[b + 1] - p0;
    const int c0 = dp.coff[b], cN = dp.coff[b + 1] - c0;
    const int x0 = dp.xoff[b];
    double *rowbuf = sm;
    double *KP = rowbuf + (RB > 0 ? (size_t)RB : (size_t)36 * nrow);          // d f / d q_k per axis (for the waypoint layer); RB: see forward_knot_body
    double *KV = KP + 3 * (nrow + 1);                 // duration adjoint of piece k, one row per axis
    double *KA = KV + 3 * (nrow + 1);
    double *Tf = KA + 3 * (nrow + 1);
    double *gT = Tf + nrow;
    double *gCo = gT + nrow;
    double *red = gCo + maxCN;
    double *xs = red + 2 * (nthr >> 6) + 2;
    double *vs = xs + maxXb;
    double *pw = vs + maxVb;                            // [nrow][nsteps*8+5] multipliers saved by the forward pass
    double *dsv = pw + (size_t)(nsteps * 8 + 5) * nrow; // [maxXb] search direction (only with a line-search tap)
    double *gs = nullptr, *gpub = nullptr;
    bool gwt = true;
    if (ro) { xs = ro->xs; vs = ro->vs; dsv = ro->dsv; pw = ro->pw; gs = ro->gs; gpub = ro->gpub; gwt = ro->gwt; }
    const int pws = nsteps * 8 + 5;
    // A resident caller that reads the host's command word while this body runs (tap.early_cmd) must find thread 0's wave WITHOUT stores of
    // its own behind that read: vmcnt retires in order and the compiler can only wait for the read with vmcnt(0), i.e. for the acknowledgement
    // of every younger store as well - measured as 1.2 us at the end of the adjoint with four clusters on an XCD, time the leader otherwise
    // spends on the line search before its next publication drains them anyway.  The time gradient's copy for the cluster (`gpub`) is
    // therefore stored by wave 1 from the LDS copy, behind the last barrier.
    const bool defer_time_gpub = gs != nullptr && gpub != nullptr && tap.early_cmd != nullptr;
    const bool tapped = tap.d != nullptr;
    const int tap_flags = (tapped && tap.flags) ? tap.flags[b] : 0;   // consumed by thread 0 at the very end
    double t_dg = 0.0, t_xx = 0.0, t_gg = 0.0;          // g.d, x.x, g.g over the elements this thread writes
    FRX_STAMP(16);
    // ---- all global reads up front ----
    const bool piece = kk < N;
    const int kp = piece ? kk : 0;                      // clamped piece index: loads are unconditional, results of lanes without a piece unused
    double h, o0 = 0.0, o1 = 0.0, c9[9], cq[6], cbq[6], r_tl[3] = {0, 0, 0};
    int r_wnv = 1, r_wvb = 0, r_wxb = 0, r_iv = 1, r_fb = 0;
    const int cv0 = __builtin_amdgcn_readfirstlane(dp.cvoff[b]);     // (up here with the other loads: see forward_knot_body)
    double2 r_wq0 = make_double2(0.0, 0.0), r_wq1 = r_wq0;   // the forward map's sums of this pair's waypoint (stage kernels: through dp.wq_glob)
    {
        const double *ci = Cin + (size_t)(p0 + kp) * 18;
        const double *o = out20 + (size_t)(p0 + kp) * 20;
        h = ct_lds ? ct_lds[kp * 19 + 18] : ldg<SH>(Tin + p0 + kp);
        const bool o_ll = SH && ro && ro->o20ll;                             // resident caller: the penalty partials arrive as granules (rk_ll_put in penalty_reduce), polled below
        if (wave == 0) {
            if (!o_ll) { o0 = ldg<SH>(o); o1 = ldg<SH>(o + 1); }
            // (the coarse-interval table of mergeToCoarseGradT too: loaded behind the barrier below, these two sat BEHIND the resident
            // caller's read of the host's command word - vmcnt retires in order - and wave 0 waited out a PCIe round trip for them)
            if (kk < cN) { r_iv = dp.coarse_iv[c0 + kk]; r_fb = dp.coarse_fbeg[c0 + kk] - p0; }
#pragma unroll
            for (int q = 0; q < 9; q++) c9[q] = ct_lds ? ct_lds[kp * 19 + 9 + q] : ldg<SH>(ci + 9 + q);
        } else {
            const int ax = wave - 1;
#pragma unroll
            for (int q = 0; q < 6; q++) { if (!o_ll) cbq[q] = ldg<SH>(o + 2 + q * 3 + ax); cq[q] = ct_lds ? ct_lds[kp * 19 + q * 3 + ax] : ldg<SH>(ci + q * 3 + ax); }
            r_tl[0] = dp.tailPVA[b * 9 + ax]; r_tl[1] = dp.tailPVA[b * 9 + 3 + ax]; r_tl[2] = dp.tailPVA[b * 9 + 6 + ax];
            if ((t2 >> 1) < N - 1) {                                // pair t2 >> 1 = waypoint
                const int gw = p0 - b + (t2 >> 1);
                r_wnv = dp.wp_nv[gw]; r_wvb = dp.wp_vbeg[gw]; r_wxb = dp.wp_xbeg[gw];
                if (!ro && dp.wq_glob) { const double2 *wq = (const double2 *)(dp.wq_glob + 4 * (size_t)gw); r_wq0 = wq[0]; r_wq1 = wq[1]; }
            }
        }
    }
    if (SH && ro && ro->o20ll) {
        // Resident caller: the 20 partials of piece kp arrive as granules tagged with the number of this evaluation; lane kk of wave 0 polls {cost, d/dT},
        // lane kk of an axis wave its six d/dc - the workgroups that integrate the penalty neither drain nor count in front of this read, and the leader
        // does not wait for them before it calls this body.  Bounded: an expired wait records RK_ERR_ARRIVE (4) in the launch's status word.
        const ll_u64 *og = ro->o20ll + 40 * (size_t)(p0 + kp);
        const unsigned tg = ro->o20tag;
        const ll_u64 t_end = (ll_u64)wall_clock64() + ro->spin_ticks;
        constexpr int NG = 6;
        ll_u64 w[NG][2];
        const int ng = wave == 0 ? 2 : 6, gbase = wave == 0 ? 0 : 2 + (wave - 1), gstep = wave == 0 ? 1 : 3;
        for (unsigned spins = 0;; spins++) {
#pragma unroll
            for (int q = 0; q < NG; q++) {
                const ll_u64 *g2 = og + 2 * (gbase + gstep * (q < ng ? q : ng - 1));
                w[q][0] = __hip_atomic_load(g2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); w[q][1] = __hip_atomic_load(g2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bool all = true;
#pragma unroll
            for (int q = 0; q < NG; q++) all = all && rk_ll_ok(w[q][0], w[q][1], tg);
            if (all) break;
            if ((spins & 31u) == 31u && (__hip_atomic_load(ro->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || (ll_u64)wall_clock64() > t_end)) {
                unsigned expect = 0u;
                __hip_atomic_compare_exchange_strong(ro->status, &expect, 4u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (wave == 0) { o0 = rk_ll_value(w[0][0], w[0][1]); o1 = rk_ll_value(w[1][0], w[1][1]); }
        else {
#pragma unroll
            for (int q = 0; q < 6; q++) cbq[q] = rk_ll_value(w[q][0], w[q][1]);
        }
    }
    if (!ro) {
        const int nx = dp.xoff[b + 1] - x0;
        const int v0 = dp.cvoff[b], nvd = 3 * (dp.cvoff[b + 1] - v0);
        stage_to_lds<4>(xs, x + x0, nx, k, nthr);
        stage_to_lds<8>(vs, dp.vrec + 3 * (size_t)v0, nvd, k, nthr);
        if (tapped) stage_to_lds<4>(dsv, tap.d + x0, nx, k, nthr);
        // saved multipliers of this candidate: one contiguous block, 16-byte loads by all threads (a single batch)
        const int ws = nsteps * 8 + 4, n2 = (N * ws) >> 1;                 // ws is even
        const double2 *src = (const double2 *)(pcrw + (size_t)p0 * ws);
        for (int i0 = k; i0 < n2; i0 += 8 * nthr) {
            double2 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i2 = i0 + u * nthr; tmp[u] = src[i2 < n2 ? i2 : n2 - 1]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i2 = i0 + u * nthr;
                if (i2 < n2) {
                    const int e = 2 * i2, kn = e / ws, ff = e - kn * ws;        // ws even: both halves belong to the same knot
                    pw[kn * (ws + 1) + ff] = tmp[u].x; pw[kn * (ws + 1) + ff + 1] = tmp[u].y;
                }
            }
        }
        __syncthreads();
    }
    FRX_STAMP(17);
    double costAcc = 0.0, gTl = 0.0;
    unsigned long long early_word = 0, early_step = 0;
    if (wave == 0) {
        // ---- jerk energy + its duration gradient (CPU.hpp:507-520, 65-75) on top of the penalty partials ----
        if (piece) {
            Tf[kk] = h;
            const double t1 = h, t2_ = t1 * t1, t3 = t2_ * t1, t4 = t2_ * t2_, t5 = t4 * t1;
            const double *c3 = c9, *c4 = c9 + 3, *c5 = c9 + 6;
            const double s33 = dot3(c3, c3), s43 = dot3(c4, c3), s44 = dot3(c4, c4), s53 = dot3(c5, c3), s54 = dot3(c5, c4), s55 = dot3(c5, c5);
            costAcc = o0 + (36.0 * s33 * t1 + 144.0 * s43 * t2_ + 192.0 * s44 * t3 + 240.0 * s53 * t3 + 720.0 * s54 * t4 + 720.0 * s55 * t5);
            gTl = o1 + (36.0 * s33 + 288.0 * s43 * t1 + 576.0 * s44 * t2_ + 720.0 * s53 * t2_ + 2880.0 * s54 * t3 + 3600.0 * s55 * t4);
        }
        // Every load of this wave has to have LANDED before the read over PCIe is issued, and no later instruction may wait on vmcnt: the
        // compiler waits for a loaded register at its first use, with vmcnt(0) when a conditional load may lie in between - for the
        // coarse-interval table that use is the merge loop behind the barrier below, and the wait took the host read's round trip with it
        // (s_waitcnt vmcnt(0) in front of the loop in the ISA, even with the values passed through an in/out asm).  The table entry
        // therefore travels through LDS: lane kk parks it in gCo[kk], which the same lane overwrites with its result.
        if (kk < cN) gCo[kk] = (double)(r_iv + (r_fb << 10));
        if (tap.early_cmd && k == 0) { early_word = __hip_atomic_load(tap.early_cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); early_step = __hip_atomic_load(tap.early_cmd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // behind the wave's first loads (see LineSearchTap)
    } else {
        const int ax = wave - 1;
        const bool act = kk >= 1 && kk <= N - 1;
        const int kc = act ? kk : 1;
        FRX_STAMP_AX(25);
        // ---- cbar = d f / d c of this axis: penalty part + jerk energy (CPU.hpp:84-92) ----
        {
            const double t1 = h, t2_ = t1 * t1, t3 = t2_ * t1, t4 = t2_ * t2_, t5 = t4 * t1;
            cbq[3] += 72.0 * cq[3] * t1 + 144.0 * cq[4] * t2_ + 240.0 * cq[5] * t3;
            cbq[4] += 144.0 * cq[3] * t2_ + 384.0 * cq[4] * t3 + 720.0 * cq[5] * t4;
            cbq[5] += 240.0 * cq[3] * t3 + 720.0 * cq[4] * t4 + 1440.0 * cq[5] * t5;
        }
        // knot states recovered from the coefficients: p = c0, v = c1, a = 2 c2 at the piece start; the next knot's from the next lane
        const double P0 = cq[0], V0 = cq[1], A0 = 2.0 * cq[2];
        double P1 = lane_down1(P0), V1 = lane_down1(V0), A1 = lane_down1(A0);
        if (kk == N - 1) { P1 = r_tl[0]; V1 = r_tl[1]; A1 = r_tl[2]; }
        // ---- Hermite adjoint of the piece; its end-of-piece parts belong to the next knot (= next lane) ----
        double db[6] = {0, 0, 0, 0, 0, 0}, hb = 0.0;
        if (piece) hermite_adjoint(h, P0, V0, A0, P1, V1, A1, cbq, db, hb);
        const double ePu = lane_up1(db[3]), eVu = lane_up1(db[4]), eAu = lane_up1(db[5]);
        double r0 = 0.0, r1 = 0.0, pbk = 0.0;
        if (act) { r0 = db[1] + eVu; r1 = db[2] + eAu; pbk = db[0] + ePu; }      // right-hand side of K mu = wbar; direct d f / d p_k (both adjacent pieces)
        FRX_STAMP_AX(26);
        // ---- mu = K^-1 wbar with the multipliers of the forward reduction (K is symmetric), neighbours by lane shifts ----
        int nst = 0;
        for (int s = 1; s < N - 1; s <<= 1) nst++;
        double muv = 0.0, mua = 0.0;                                   // zero at the fixed end knots
        if (nst <= 6) {
            // every multiplier of this knot up front (they were all saved by the forward pass): a step is then one lane exchange, not two LDS round trips
            double ab[6][8], Di[4];
#pragma unroll
            for (int st = 0; st < 6; st++)
#pragma unroll
                for (int i = 0; i < 8; i++) ab[st][i] = pw[kc * pws + (st < nst ? st : 0) * 8 + i];
#pragma unroll
            for (int i = 0; i < 4; i++) Di[i] = pw[kc * pws + nsteps * 8 + i];
#pragma unroll
            for (int st = 0; st < 6; st++)
                if (st < nst) {
                    const int s = 1 << st;
                    const bool inlo = act && kk - s >= 1, inhi = act && kk + s <= N - 1;
                    const double l0r = __shfl_up(r0, s, 64), l1r = __shfl_up(r1, s, 64), h0r = __shfl_down(r0, s, 64), h1r = __shfl_down(r1, s, 64);
                    const double l0 = inlo ? l0r : 0.0, l1 = inlo ? l1r : 0.0, h0 = inhi ? h0r : 0.0, h1 = inhi ? h1r : 0.0;
                    const double n0 = r0 - (ab[st][0] * l0 + ab[st][1] * l1) - (ab[st][4] * h0 + ab[st][5] * h1);    // pcr_rhs_step
                    const double n1 = r1 - (ab[st][2] * l0 + ab[st][3] * l1) - (ab[st][6] * h0 + ab[st][7] * h1);
                    r0 = n0; r1 = n1;
                }
            if (act) { muv = Di[0] * r0 + Di[1] * r1; mua = Di[2] * r0 + Di[3] * r1; }
        } else {
            for (int st = 0; st < nst; st++) {
                const int s = 1 << st;
                const bool inlo = act && kk - s >= 1, inhi = act && kk + s <= N - 1;
                double ab[8];
#pragma unroll
                for (int i = 0; i < 8; i++) ab[i] = pw[kc * pws + st * 8 + i];
                const double l0r = __shfl_up(r0, s, 64), l1r = __shfl_up(r1, s, 64), h0r = __shfl_down(r0, s, 64), h1r = __shfl_down(r1, s, 64);
                const double l0 = inlo ? l0r : 0.0, l1 = inlo ? l1r : 0.0, h0 = inhi ? h0r : 0.0, h1 = inhi ? h1r : 0.0;
                const double n0 = r0 - (ab[0] * l0 + ab[1] * l1) - (ab[4] * h0 + ab[5] * h1);
                const double n1 = r1 - (ab[2] * l0 + ab[3] * l1) - (ab[6] * h0 + ab[7] * h1);
                r0 = n0; r1 = n1;
            }
            if (act) {
                const double *Di = pw + kk * pws + nsteps * 8;
                muv = Di[0] * r0 + Di[1] * r1; mua = Di[2] * r0 + Di[3] * r1;
            }
        }
        FRX_STAMP_AX(27);
        // ---- through the knot system: duration term and d f / d(p_{k+1} - p_k) ----
        double mu1v = lane_down1(muv), mu1a = lane_down1(mua);
        if (kk >= N - 1) { mu1v = 0.0; mu1a = 0.0; }
        double dlb = 0.0;
        if (piece) dlb = knot_adjoint_piece(h, P1 - P0, V0, A0, V1, A1, muv, mua, mu1v, mu1a, hb);
        const double dlu = lane_up1(dlb);                    // + dl of the piece ending at this knot
        if (piece) KN(KV, ax, kk) = hb;
        if (act) KN(KP, ax, kk) = pbk + dlu - dlb;                   // d f / d q_k for the pair that owns the waypoint
        FRX_STAMP_AX(28);
    }
    __syncthreads();
    FRX_STAMP(22);
    if (wave == 0) {
        // ---- duration gradient, mergeToCoarseGradT (CPU.hpp:946-959), cost (CPU.hpp:988), addLayerTGrad (CPU.hpp:816-894): all within wave 0 ----
        if (piece) gT[kk] = gTl + ((KN(KV, 0, kk) + KN(KV, 1, kk)) + KN(KV, 2, kk)) + dp.rho;     // + rho: CPU.hpp:989
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double sumTc = 0.0;
        if (kk < cN) {
            const int packed = (int)gCo[kk], iv = packed & 1023, fb = packed >> 10;   // (parked above: see the note at the host read)
            double sg = 0.0, tt = 0.0;
            for (int a = 0; a < iv; a++) { sg += gT[fb + a]; tt += Tf[fb + a]; }
            gCo[kk] = sg / iv;
            sumTc = tt;
        }
        const double wc = wave_sum_dpp(costAcc), wtt = wave_sum_dpp(sumTc);
        const double fval = wc + dp.rho * wtt;
        if (kk == 0) { if (!ro) f[b] = fval; red[0] = fval; }             // resident caller: the value travels through the mailbox, nothing to drain
        if (dp.soft) {
            if (kk < cN) {
                const double gi = gCo[kk] * dT_dtau(xs[kk], dp.c2 != 0);
                if (gs) gs[kk] = gi; else g[x0 + kk] = gi;
                if (gpub && !defer_time_gpub) stg<SH>(gpub + kk, gi, gwt);
                if (tapped) { t_dg += gi * dsv[kk]; t_xx += xs[kk] * xs[kk]; t_gg += gi * gi; }
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (kk == 0) {
                const int Ms1 = cN - 1;
                const double gTail = dp.sumT * gCo[Ms1];
                double expTauSum = 0.0, gFreeDotExpTau = 0.0;
                for (int i = 0; i < Ms1; i++) {
                    const double e = tau_to_T(xs[i], dp.c2 != 0);
                    expTauSum += e;
                    gFreeDotExpTau += e * (dp.sumT * gCo[i]);
                }
                const double den = expTauSum + 1.0;
                for (int i = 0; i < Ms1; i++) {
                    const double de = dT_dtau(xs[i], dp.c2 != 0);
                    const double gi = (dp.sumT * gCo[i] - gTail) * de / den - (gFreeDotExpTau - gTail * expTauSum) * de / (den * den);
                    if (gs) gs[i] = gi; else g[x0 + i] = gi;
                    if (gpub && !defer_time_gpub) stg<SH>(gpub + i, gi, gwt);
                    if (tapped) { t_dg += gi * dsv[i]; t_xx += xs[i] * xs[i]; t_gg += gi * gi; }
                }
            }
        }
    } else {
        // ---- addPropCtoP + addLayerPGrad (CPU.hpp:154-161, 897-928): waypoint w (= knot w+1) on a PAIR of lanes of the axis waves ----
        for (int w0 = 0; w0 < N - 1; w0 += 96) {
            const int w = w0 + (t2 >> 1), sub = t2 & 1;
            const bool wact = w < N - 1;
            const double *V = vs, *xi = xs;
            int nv1 = 0, xb = 0;
            double g0 = 0.0, g1 = 0.0, g2 = 0.0, qn = 0.0;
            if (wact) {
                int wnv = r_wnv, wvb = r_wvb, wxb = r_wxb;
                if (w0 > 0) { const int gw = p0 - b + w; wnv = dp.wp_nv[gw]; wvb = dp.wp_vbeg[gw]; wxb = dp.wp_xbeg[gw]; }
                nv1 = wnv - 1; xb = wxb;
                V = vs + 3 * (wvb - cv0) + (ro ? ro->vskew * w : 0);
                xi = xs + (xb - x0);
                g0 = KN(KP, 0, w + 1); g1 = KN(KP, 1, w + 1); g2 = KN(KP, 2, w + 1);
            }
            // with r_a = sc xi_a, sc = 2 / (1 + |xi|^2):  d f / d xi_a = xi_a (2 sc^2 (V_a . g) - 4 gdq / (1 + |xi|^2)^2),
            // gdq = 2 sc sum_a (V_a . g) xi_a^2  -  so |xi|^2 and the weighted sum come out of ONE pass over the vertices
            double s2 = 0.0;
            // the forward map of this evaluation left both sums behind: in LDS (same workgroup, resident caller) or in dp.wq_glob (stage kernels; first pass only)
            const bool cached_lds = ro && ro->wq, cached = cached_lds || (!ro && dp.wq_glob && w0 == 0);
            if (wact && cached_lds) { const double *wq = ro->wq + 4 * w; qn = wq[0]; s2 = wq[1] * g0 + wq[2] * g1 + wq[3] * g2; }
            else if (wact && cached) { qn = r_wq0.x; s2 = r_wq0.y * g0 + r_wq1.x * g1 + r_wq1.y * g2; }
            if (wact && !cached)
                for (int a0 = sub; a0 < nv1; a0 += 8) {                   // four vertices per trip, their LDS reads in flight together (clamped, not predicated)
                    double xv[4], dgv[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { const int a = min(a0 + 2 * j, nv1 - 1); xv[j] = xi[a]; dgv[j] = V[3 * (a + 1)] * g0 + V[3 * (a + 1) + 1] * g1 + V[3 * (a + 1) + 2] * g2; }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (a0 + 2 * j < nv1) { const double x2 = xv[j] * xv[j]; qn += x2; s2 += dgv[j] * x2; }
                }
            FRX_STAMP_AX(30);
            if (!cached) { qn += dpp_mov<0xB1>(qn); s2 += dpp_mov<0xB1>(s2); }   // pair sums
            const double qp1 = qn + 1.0, iq = 1.0 / qp1, sc = 2.0 * iq;
            const double gdq = 2.0 * sc * s2, kq = 4.0 * gdq * (iq * iq), sc22 = 2.0 * sc * sc;
            FRX_STAMP_AX(31);
            if (wact)
                for (int a0 = sub; a0 < nv1; a0 += 8) {
                    double xv[4], dgv[4], dd[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int a = min(a0 + 2 * j, nv1 - 1);
                        xv[j] = xi[a]; dgv[j] = V[3 * (a + 1)] * g0 + V[3 * (a + 1) + 1] * g1 + V[3 * (a + 1) + 2] * g2;
                        dd[j] = tapped ? dsv[xb - x0 + a] : 0.0;
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (a0 + 2 * j < nv1) {
                            const double gi = xv[j] * (sc22 * dgv[j] - kq);
                            if (gs) gs[xb - x0 + a0 + 2 * j] = gi; else g[xb + a0 + 2 * j] = gi;
                            if (gpub) stg<SH>(gpub + (xb - x0 + a0 + 2 * j), gi, gwt);
                            t_dg += gi * dd[j]; t_xx += xv[j] * xv[j]; t_gg += gi * gi;
                        }
                }
        }
    }
    FRX_STAMP_AX(29);
    FRX_STAMP(23);
    // ---- line-search tap: what lbfgs.hpp:830 (g.d) and :1296-1297 (|x|, |g|) need, reduced here instead of in a separate launch ----
    if (tap.d != nullptr) {                                           // uniform over the grid
        const double w0 = wave_sum_dpp(t_dg), w1 = wave_sum_dpp(t_xx), w2 = wave_sum_dpp(t_gg);
        const int nw = nthr >> 6, w = k >> 6;
        double *red3 = rowbuf;                                        // the row buffer is not used by this path
        if ((k & 63) == 0) { red3[w] = w0; red3[nw + w] = w1; red3[2 * nw + w] = w2; }
        __syncthreads();
        if (defer_time_gpub && wave == 1 && kk < (dp.soft ? cN : cN - 1)) stg<SH>(gpub + kk, gs[kk], gwt);   // (see defer_time_gpub)
        if (k == 0 && ((tap_flags & DV_EVAL) || tap.lds_out)) {
            const double fval = red[0];
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (int i = 0; i < nw; i++) { a0 += red3[i]; a1 += red3[nw + i]; a2 += red3[2 * nw + i]; }
            if (tap.lds_out) { tap.lds_out[0] = fval; tap.lds_out[1] = a0; tap.lds_out[2] = a1; tap.lds_out[3] = a2; if (tap.early_cmd) { tap.lds_out[7] = __longlong_as_double((long long)early_word); tap.lds_out[6] = __longlong_as_double((long long)early_step); } }
            else { DvResult *r = tap.res + b; r->f = fval; r->dg = a0; r->xx = a1; r->gg = a2; }
        }
        if (k == 0 && tap.arrive) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // system scope: the result above is visible before the count moves
            if (atomicAdd(tap.arrive, 1u) + 1u == (unsigned)gridDim.x * tap.round) *tap.flag = tap.round;
        }
    }
    FRX_STAMP(24);
}

template <bool SH>
__device__ __forceinline__ void backward_knot_wsp64(const DevProblem &dp, const double *__restrict__ x, const double *__restrict__ Tin,
                                const double *__restrict__ Cin, const double *__restrict__ out20, double *__restrict__ f,
                                double *__restrict__ g, int maxCN, int maxXb, int maxVb, const double *__restrict__ pcrw, int nsteps,
                                const LineSearchTap &tap, int b, double *sm, const double *ct_lds, const ResidentOps *ro) {
    const int nrow = 64, nthr = 256;
    const int k = threadIdx.x, kk = k & 63, t2 = k - 64;
    const int wave = __builtin_amdgcn_readfirstlane(k >> 6);
    const int p0 = dp.poff[b], N = dp.poff[b + 1] - p0;
    const int c0 = dp.coff[b], cN = dp.coff[b + 1] - c0;
    const int x0 = dp.xoff[b];
    double *rowbuf = sm;
    double *KP = rowbuf + (size_t)36 * nrow;          // d f / d q_k per axis (for the waypoint layer)
    double *KV = KP + 3 * (nrow + 1);                 // duration adjoint of piece k, one row per axis
    double *KA = KV + 3 * (nrow + 1);
    double *Tf = KA + 3 * (nrow + 1);
    double *gT = Tf + nrow;
    double *gCo = gT + nrow;
    double *red = gCo + maxCN;
    double *xs = red + 2 * (nthr >> 6) + 2;
    double *vs = xs + maxXb;
    double *pw = vs + maxVb;                            // [nrow][nsteps*8+5] multipliers saved by the forward pass
    double *dsv = pw + (size_t)(nsteps * 8 + 5) * nrow; // [maxXb] search direction (only with a line-search tap)
    double *gs = nullptr, *gpub = nullptr;
    bool gwt = true;
    if (ro) { xs = ro->xs; vs = ro->vs; dsv = ro->dsv; pw = ro->pw; gs = ro->gs; gpub = ro->gpub; gwt = ro->gwt; }
    const int pws = nsteps * 8 + 5;
    // A resident caller that reads the host's command word while this body runs (tap.early_cmd) must find thread 0's wave WITHOUT stores of
    // its own behind that read: vmcnt retires in order and the compiler can only wait for the read with vmcnt(0), i.e. for the acknowledgement
    // of every younger store as well - measured as 1.2 us at the end of the adjoint with four clusters on an XCD, time the leader otherwise
    // spends on the line search before its next publication drains them anyway.  The time gradient's copy for the cluster (`gpub`) is
    // therefore stored by wave 1 from the LDS copy, behind the last barrier.
    const bool defer_time_gpub = gs != nullptr && gpub != nullptr && tap.early_cmd != nullptr;
    const bool tapped = tap.d != nullptr;
    const int tap_flags = (tapped && tap.flags) ? tap.flags[b] : 0;   // consumed by thread 0 at the very end
    double t_dg = 0.0, t_xx = 0.0, t_gg = 0.0;          // g.d, x.x, g.g over the elements this thread writes
    FRX_STAMP(16);
    // ---- all global reads up front ----
    const bool piece = kk < N;
    const int kp = piece ? kk : 0;                      // clamped piece index: loads are unconditional, results of lanes without a piece unused
    double h, o0 = 0.0, o1 = 0.0, c9[9], cq[6], cbq[6], r_tl[3] = {0, 0, 0};
    int r_wnv = 1, r_wvb = 0, r_wxb = 0, r_iv = 1, r_fb = 0;
    const int cv0 = __builtin_amdgcn_readfirstlane(dp.cvoff[b]);     // (up here with the other loads: see forward_knot_body)
    double2 r_wq0 = make_double2(0.0, 0.0), r_wq1 = r_wq0;   // the forward map's sums of this pair's waypoint (stage kernels: through dp.wq_glob)
    {
        const double *ci = Cin + (size_t)(p0 + kp) * 18;
        const double *o = out20 + (size_t)(p0 + kp) * 20;
        h = ct_lds ? ct_lds[kp * 19 + 18] : ldg<SH>(Tin + p0 + kp);
        const bool o_ll = SH && ro && ro->o20ll;                             // resident caller: the penalty partials arrive as granules (rk_ll_put in penalty_reduce), polled below
        if (wave == 0) {
            if (!o_ll) { o0 = ldg<SH>(o); o1 = ldg<SH>(o + 1); }
            // (the coarse-interval table of mergeToCoarseGradT too: loaded behind the barrier below, these two sat BEHIND the resident
            // caller's read of the host's command word - vmcnt retires in order - and wave 0 waited out a PCIe round trip for them)
            if (kk < cN) { r_iv = dp.coarse_iv[c0 + kk]; r_fb = dp.coarse_fbeg[c0 + kk] - p0; }
#pragma unroll
            for (int q = 0; q < 9; q++) c9[q] = ct_lds ? ct_lds[kp * 19 + 9 + q] : ldg<SH>(ci + 9 + q);
        } else {
            const int ax = wave - 1;
#pragma unroll
            for (int q = 0; q < 6; q++) { if (!o_ll) cbq[q] = ldg<SH>(o + 2 + q * 3 + ax); cq[q] = ct_lds ? ct_lds[kp * 19 + q * 3 + ax] : ldg<SH>(ci + q * 3 + ax); }
            r_tl[0] = dp.tailPVA[b * 9 + ax]; r_tl[1] = dp.tailPVA[b * 9 + 3 + ax]; r_tl[2] = dp.tailPVA[b * 9 + 6 + ax];
            if ((t2 >> 1) < N - 1) {                                // pair t2 >> 1 = waypoint
                const int gw = p0 - b + (t2 >> 1);
                r_wnv = dp.wp_nv[gw]; r_wvb = dp.wp_vbeg[gw]; r_wxb = dp.wp_xbeg[gw];
                if (!ro && dp.wq_glob) { const double2 *wq = (const double2 *)(dp.wq_glob + 4 * (size_t)gw); r_wq0 = wq[0]; r_wq1 = wq[1]; }
            }
        }
    }
    if (!ro) {
        const int nx = dp.xoff[b + 1] - x0;
        const int v0 = dp.cvoff[b], nvd = 3 * (dp.cvoff[b + 1] - v0);
        stage_to_lds<4>(xs, x + x0, nx, k, nthr);
        stage_to_lds<8>(vs, dp.vrec + 3 * (size_t)v0, nvd, k, nthr);
        if (tapped) stage_to_lds<4>(dsv, tap.d + x0, nx, k, nthr);
        // saved multipliers of this candidate: one contiguous block, 16-byte loads by all threads (a single batch)
        const int ws = nsteps * 8 + 4, n2 = (N * ws) >> 1;                 // ws is even
        const double2 *src = (const double2 *)(pcrw + (size_t)p0 * ws);
        for (int i0 = k; i0 < n2; i0 += 8 * nthr) {
            double2 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i2 = i0 + u * nthr; tmp[u] = src[i2 < n2 ? i2 : n2 - 1]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i2 = i0 + u * nthr;
                if (i2 < n2) {
                    const int e = 2 * i2, kn = e / ws, ff = e - kn * ws;        // ws even: both halves belong to the same knot
                    pw[kn * (ws + 1) + ff] = tmp[u].x; pw[kn * (ws + 1) + ff + 1] = tmp[u].y;
                }
            }
        }
        __syncthreads();
    }
    FRX_STAMP(17);
    // ================================================================================================================================================
    // Round 5: everything that does NOT depend on the penalty partials runs here, IN FRONT of the poll for them.  In the resident kernel the leader has
    // ~4.5 us between the end of its forward map and the arrival of the partials (hand-off, the members' penalty share) with nothing to do; until
    // round 4 it then ran, behind the poll, a chain whose first third did not need them: durations' powers and reciprocals, the Hermite adjoint's and the
    // knot adjoint's coefficients (functions of h, p, v, a alone), the 52 saved multipliers of its knot (LDS, three axis waves sharing the read
    // bandwidth), the waypoint layer's vertices, variables and direction elements, x.x, the tau layer's derivative.  The compiler cannot move any of
    // it across the polling loop.  Same expressions, same order of operations on the values that do depend on the partials: bit-identical results.
    // (The stage kernels have no poll: for them this is the order the scheduler chose anyway.)
    // ================================================================================================================================================
    double costAcc = 0.0, gTl = 0.0;
    unsigned long long early_word = 0, early_step = 0;
    int nst = 0;
    for (int s = 1; s < N - 1; s <<= 1) nst++;
    const int ax = wave - 1;                                          // (axis waves)
    const bool act = kk >= 1 && kk <= N - 1;
    const int kc = act ? kk : 1;
    // wave 0
    double jerkE = 0.0, jerkT = 0.0, dtt = 0.0, x_tau = 0.0, d_tau = 0.0;
    // axis waves: Hermite adjoint / knot adjoint coefficients of piece kk, multipliers of knot kk
    double cj3 = 0.0, cj4 = 0.0, cj5 = 0.0, P0 = 0.0, V0 = 0.0, A0 = 0.0, P1 = 0.0, V1 = 0.0, A1 = 0.0;
    double hk_pd3 = 0, hk_pd4 = 0, hk_pd5 = 0, hk_13 = 0, hk_14 = 0, hk_15 = 0, hk_43 = 0, hk_44 = 0, hk_23 = 0, hk_24 = 0, hk_25 = 0, hk_53 = 0, hk_54 = 0, hk_55 = 0, hk_dc3 = 0, hk_dc4 = 0, hk_dc5 = 0;
    double ka_dLv = 0, ka_dLa = 0, ka_dRv = 0, ka_dRa = 0, ka_ih3 = 0, ka_ih4 = 0;
    double ab[6][8], Di[4] = {0, 0, 0, 0};
    // axis waves, waypoint layer: this lane's vertices of its pair's waypoint (two trips of four, as the loop below takes them)
    constexpr int WPF = 8;
    const int wpt = t2 >> 1, sub = t2 & 1;
    const bool wact0 = wave >= 1 && wpt < N - 1;
    const bool cached_lds = ro && ro->wq, cached0 = SH || cached_lds || (!ro && dp.wq_glob != nullptr);   // (SH: the resident caller always hands the forward map's sums over - the two-pass form below is not compiled into it)
    const double *Vw = vs, *xiw = xs;
    int nv1w = 0, xbw = 0;
    double qn0 = 0.0, wq1 = 0.0, wq2 = 0.0, wq3 = 0.0, w_c2sc = 0.0, w_iq2 = 0.0, w_sc22 = 0.0;
    double pxv[WPF], pvx[WPF], pvy[WPF], pvz[WPF], pdd[WPF];
    if (wave == 0) {
        // ---- jerk energy + its duration gradient (CPU.hpp:507-520, 65-75): the part that is added to the penalty partials ----
        if (piece) {
            Tf[kk] = h;
            const double t1 = h, t2_ = t1 * t1, t3 = t2_ * t1, t4 = t2_ * t2_, t5 = t4 * t1;
            const double *c3 = c9, *c4 = c9 + 3, *c5 = c9 + 6;
            const double s33 = dot3(c3, c3), s43 = dot3(c4, c3), s44 = dot3(c4, c4), s53 = dot3(c5, c3), s54 = dot3(c5, c4), s55 = dot3(c5, c5);
            jerkE = (36.0 * s33 * t1 + 144.0 * s43 * t2_ + 192.0 * s44 * t3 + 240.0 * s53 * t3 + 720.0 * s54 * t4 + 720.0 * s55 * t5);
            jerkT = (36.0 * s33 + 288.0 * s43 * t1 + 576.0 * s44 * t2_ + 720.0 * s53 * t2_ + 2880.0 * s54 * t3 + 3600.0 * s55 * t4);
        }
        if (dp.soft && kk < cN) { x_tau = xs[kk]; dtt = dT_dtau(x_tau, dp.c2 != 0); if (tapped) d_tau = dsv[kk]; }
    } else {
        FRX_STAMP_AX(25);
        {   // jerk-energy part of cbar = d f / d c of this axis (CPU.hpp:84-92)
            const double t1 = h, t2_ = t1 * t1, t3 = t2_ * t1, t4 = t2_ * t2_, t5 = t4 * t1;
            cj3 = 72.0 * cq[3] * t1 + 144.0 * cq[4] * t2_ + 240.0 * cq[5] * t3;
            cj4 = 144.0 * cq[3] * t2_ + 384.0 * cq[4] * t3 + 720.0 * cq[5] * t4;
            cj5 = 240.0 * cq[3] * t3 + 720.0 * cq[4] * t4 + 1440.0 * cq[5] * t5;
        }
        // knot states recovered from the coefficients: p = c0, v = c1, a = 2 c2 at the piece start; the next knot's from the next lane
        P0 = cq[0]; V0 = cq[1]; A0 = 2.0 * cq[2];
        P1 = lane_down1(P0); V1 = lane_down1(V0); A1 = lane_down1(A0);
        if (kk == N - 1) { P1 = r_tl[0]; V1 = r_tl[1]; A1 = r_tl[2]; }
        {   // hermite_adjoint (frx_minco.hpp), the factors of cbar: term by term the sub-expressions of that function
            const double ih = rcp_fast(h), ih2 = ih * ih, ih3 = ih2 * ih, ih4 = ih2 * ih2, ih5 = ih4 * ih, ih6 = ih3 * ih3;
            const double dl = P1 - P0;
            hk_pd3 = 10.0 * ih3; hk_pd4 = 15.0 * ih4; hk_pd5 = 6.0 * ih5;
            hk_13 = 6.0 * ih2; hk_14 = 8.0 * ih3; hk_15 = 3.0 * ih4;
            hk_43 = -4.0 * ih2; hk_44 = 7.0 * ih3;
            hk_23 = 1.5 * ih; hk_24 = 1.5 * ih2; hk_25 = 0.5 * ih3;
            hk_53 = 0.5 * ih; hk_54 = ih2; hk_55 = 0.5 * ih3;
            hk_dc3 = -30.0 * dl * ih4 + 2.0 * (4.0 * V1 + 6.0 * V0) * ih3 + 0.5 * (3.0 * A0 - A1) * ih2;
            hk_dc4 = 60.0 * dl * ih5 - 3.0 * (7.0 * V1 + 8.0 * V0) * ih4 - (3.0 * A0 - 2.0 * A1) * ih3;
            hk_dc5 = -30.0 * dl * ih6 + 12.0 * (V1 + V0) * ih5 + 1.5 * (A0 - A1) * ih4;
            // knot_adjoint_piece (frx_minco.hpp): the rows this piece contributes to its two knots
            ka_dLv = -504.0 * V0 * ih4 - 48.0 * A0 * ih3 - 576.0 * V1 * ih4 + 72.0 * A1 * ih3 + 1440.0 * dl * ih5;
            ka_dLa = 48.0 * V0 * ih3 + 3.0 * A0 * ih2 + 72.0 * V1 * ih3 - 9.0 * A1 * ih2 - 180.0 * dl * ih4;
            ka_dRv = 1440.0 * dl * ih5 - 576.0 * V0 * ih4 - 504.0 * V1 * ih4 - 72.0 * A0 * ih3 + 48.0 * A1 * ih3;
            ka_dRa = 180.0 * dl * ih4 - 72.0 * V0 * ih3 - 48.0 * V1 * ih3 - 9.0 * A0 * ih2 + 3.0 * A1 * ih2;
            ka_ih3 = ih3; ka_ih4 = ih4;
        }
    }
    if (SH && ro && ro->o20ll) {
        // Resident caller: the 20 partials of piece kp arrive as granules tagged with the number of this evaluation; lane kk of wave 0 polls {cost, d/dT},
        // lane kk of an axis wave its six d/dc - the workgroups that integrate the penalty neither drain nor count in front of this read, and the leader
        // does not wait for them before it calls this body.  Bounded: an expired wait records RK_ERR_ARRIVE (4) in the launch's status word.
        const ll_u64 *og = ro->o20ll + 40 * (size_t)(p0 + kp);
        const unsigned tg = ro->o20tag;
        const ll_u64 t_end = (ll_u64)wall_clock64() + ro->spin_ticks;
        constexpr int NG = 6;
        ll_u64 w[NG][2];
        const int ng = wave == 0 ? 2 : 6, gbase = wave == 0 ? 0 : 2 + (wave - 1), gstep = wave == 0 ? 1 : 3;
        for (unsigned spins = 0;; spins++) {
#pragma unroll
            for (int q = 0; q < NG; q++) {
                const ll_u64 *g2 = og + 2 * (gbase + gstep * (q < ng ? q : ng - 1));
                w[q][0] = __hip_atomic_load(g2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); w[q][1] = __hip_atomic_load(g2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bool all = true;
#pragma unroll
            for (int q = 0; q < NG; q++) all = all && rk_ll_ok(w[q][0], w[q][1], tg);
            if (all) break;
            if ((spins & 31u) == 31u && (__hip_atomic_load(ro->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || (ll_u64)wall_clock64() > t_end)) {
                unsigned expect = 0u;
                __hip_atomic_compare_exchange_strong(ro->status, &expect, 4u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (wave == 0) { o0 = rk_ll_value(w[0][0], w[0][1]); o1 = rk_ll_value(w[1][0], w[1][1]); }
        else {
#pragma unroll
            for (int q = 0; q < 6; q++) cbq[q] = rk_ll_value(w[q][0], w[q][1]);
        }
    }
    if (wave == 0) {
        if (piece) { costAcc = o0 + jerkE; gTl = o1 + jerkT; }
        // Every load of this wave has to have LANDED before the read over PCIe is issued, and no later instruction may wait on vmcnt: the
        // compiler waits for a loaded register at its first use, with vmcnt(0) when a conditional load may lie in between - for the
        // coarse-interval table that use is the merge loop behind the barrier below, and the wait took the host read's round trip with it
        // (s_waitcnt vmcnt(0) in front of the loop in the ISA, even with the values passed through an in/out asm).  The table entry
        // therefore travels through LDS: lane kk parks it in gCo[kk], which the same lane overwrites with its result.
        if (kk < cN) gCo[kk] = (double)(r_iv + (r_fb << 10));
        if (tap.early_cmd && k == 0) { early_word = __hip_atomic_load(tap.early_cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); early_step = __hip_atomic_load(tap.early_cmd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // behind the wave's first loads (see LineSearchTap)
    } else {
        if (nst <= 6) {
            // every multiplier of this knot (they were all saved by the forward pass), requested first: a step of the solve below is then one lane exchange,
            // not two LDS round trips, and these trips run under the Hermite adjoint.  (In front of the poll they would be 104 more live registers there.)
#pragma unroll
            for (int st = 0; st < 6; st++)
#pragma unroll
                for (int i = 0; i < 8; i++) ab[st][i] = pw[kc * pws + (st < nst ? st : 0) * 8 + i];
#pragma unroll
            for (int i = 0; i < 4; i++) Di[i] = pw[kc * pws + nsteps * 8 + i];
        }
        // ---- cbar = d f / d c of this axis: penalty part + jerk energy (CPU.hpp:84-92) ----
        cbq[3] += cj3; cbq[4] += cj4; cbq[5] += cj5;
        // ---- Hermite adjoint of the piece (hermite_adjoint, coefficients from above); its end-of-piece parts belong to the next knot (= next lane) ----
        double db[6] = {0, 0, 0, 0, 0, 0}, hb = 0.0;
        if (piece) {
            const double *cb = cbq;
            const double pd = hk_pd3 * cb[3] - hk_pd4 * cb[4] + hk_pd5 * cb[5];
            db[0] = cb[0] - pd;
            db[3] = pd;
            db[1] = cb[1] - hk_13 * cb[3] + hk_14 * cb[4] - hk_15 * cb[5];
            db[4] = hk_43 * cb[3] + hk_44 * cb[4] - hk_15 * cb[5];
            db[2] = 0.5 * cb[2] - hk_23 * cb[3] + hk_24 * cb[4] - hk_25 * cb[5];
            db[5] = hk_53 * cb[3] - hk_54 * cb[4] + hk_55 * cb[5];
            hb = cb[3] * hk_dc3 + cb[4] * hk_dc4 + cb[5] * hk_dc5;
        }
        const double ePu = lane_up1(db[3]), eVu = lane_up1(db[4]), eAu = lane_up1(db[5]);
        double r0 = 0.0, r1 = 0.0, pbk = 0.0;
        if (act) { r0 = db[1] + eVu; r1 = db[2] + eAu; pbk = db[0] + ePu; }      // right-hand side of K mu = wbar; direct d f / d p_k (both adjacent pieces)
        FRX_STAMP_AX(26);
        // ---- mu = K^-1 wbar with the multipliers of the forward reduction (K is symmetric), neighbours by lane shifts ----
        double muv = 0.0, mua = 0.0;                                   // zero at the fixed end knots
        if (nst <= 6) {
#pragma unroll
            for (int st = 0; st < 6; st++)
                if (st < nst) {
                    const int s = 1 << st;
                    const bool inlo = act && kk - s >= 1, inhi = act && kk + s <= N - 1;
                    const double l0r = __shfl_up(r0, s, 64), l1r = __shfl_up(r1, s, 64), h0r = __shfl_down(r0, s, 64), h1r = __shfl_down(r1, s, 64);
                    const double l0 = inlo ? l0r : 0.0, l1 = inlo ? l1r : 0.0, h0 = inhi ? h0r : 0.0, h1 = inhi ? h1r : 0.0;
                    const double n0 = r0 - (ab[st][0] * l0 + ab[st][1] * l1) - (ab[st][4] * h0 + ab[st][5] * h1);    // pcr_rhs_step
                    const double n1 = r1 - (ab[st][2] * l0 + ab[st][3] * l1) - (ab[st][6] * h0 + ab[st][7] * h1);
                    r0 = n0; r1 = n1;
                }
            if (act) { muv = Di[0] * r0 + Di[1] * r1; mua = Di[2] * r0 + Di[3] * r1; }
        } else {
            for (int st = 0; st < nst; st++) {
                const int s = 1 << st;
                const bool inlo = act && kk - s >= 1, inhi = act && kk + s <= N - 1;
                double ab1[8];
#pragma unroll
                for (int i = 0; i < 8; i++) ab1[i] = pw[kc * pws + st * 8 + i];
                const double l0r = __shfl_up(r0, s, 64), l1r = __shfl_up(r1, s, 64), h0r = __shfl_down(r0, s, 64), h1r = __shfl_down(r1, s, 64);
                const double l0 = inlo ? l0r : 0.0, l1 = inlo ? l1r : 0.0, h0 = inhi ? h0r : 0.0, h1 = inhi ? h1r : 0.0;
                const double n0 = r0 - (ab1[0] * l0 + ab1[1] * l1) - (ab1[4] * h0 + ab1[5] * h1);
                const double n1 = r1 - (ab1[2] * l0 + ab1[3] * l1) - (ab1[6] * h0 + ab1[7] * h1);
                r0 = n0; r1 = n1;
            }
            if (act) {
                const double *Dq = pw + kk * pws + nsteps * 8;
                muv = Dq[0] * r0 + Dq[1] * r1; mua = Dq[2] * r0 + Dq[3] * r1;
            }
        }
        FRX_STAMP_AX(27);
        // The waypoint layer's operands - this lane's vertices, variables and direction elements, the forward map's two sums - are requested HERE, where the
        // solve's 52 multipliers are dead: their LDS round trips run under the knot adjoint and the barrier instead of behind it (in front of the
        // poll, next to the multipliers, they cost the resident kernel's leader 190 spilled registers).
        if (wact0) {                                                  // (<= 64 pieces: 63 waypoints on the 96 lane pairs of the axis waves - one pass)
            nv1w = r_wnv - 1; xbw = r_wxb;
            Vw = vs + 3 * (r_wvb - cv0) + (ro ? ro->vskew * wpt : 0);
            xiw = xs + (xbw - x0);
            if (cached_lds) { const double *wq = ro->wq + 4 * wpt; qn0 = wq[0]; wq1 = wq[1]; wq2 = wq[2]; wq3 = wq[3]; }
            else if (cached0) { qn0 = r_wq0.x; wq1 = r_wq0.y; wq2 = r_wq1.x; wq3 = r_wq1.y; }
#pragma unroll
            for (int j = 0; j < WPF; j++) {
                const int a = min(sub + 2 * j, nv1w - 1);
                pxv[j] = xiw[a]; pvx[j] = Vw[3 * (a + 1)]; pvy[j] = Vw[3 * (a + 1) + 1]; pvz[j] = Vw[3 * (a + 1) + 2];
                pdd[j] = tapped ? dsv[xbw - x0 + a] : 0.0;
            }
            if (cached0) {
                const double qp1 = qn0 + 1.0, iq = 1.0 / qp1, sc = 2.0 * iq;
                w_c2sc = 2.0 * sc; w_iq2 = iq * iq; w_sc22 = 2.0 * sc * sc;
#pragma unroll
                for (int j = 0; j < WPF; j++) if (sub + 2 * j < nv1w) t_xx += pxv[j] * pxv[j];
            }
        }
        // ---- through the knot system (knot_adjoint_piece, rows from above): duration term and d f / d(p_{k+1} - p_k) ----
        double mu1v = lane_down1(muv), mu1a = lane_down1(mua);
        if (kk >= N - 1) { mu1v = 0.0; mu1a = 0.0; }
        double dlb = 0.0;
        if (piece) {
            hb -= mu1v * ka_dLv + mu1a * ka_dLa + muv * ka_dRv + mua * ka_dRa;
            dlb = (mu1v + muv) * 360.0 * ka_ih4 + (mua - mu1a) * 60.0 * ka_ih3;
        }
        const double dlu = lane_up1(dlb);                    // + dl of the piece ending at this knot
        if (piece) KN(KV, ax, kk) = hb;
        if (act) KN(KP, ax, kk) = pbk + dlu - dlb;                   // d f / d q_k for the pair that owns the waypoint
        FRX_STAMP_AX(28);
    }
    __syncthreads();
    FRX_STAMP(22);
    if (wave == 0) {
        // ---- duration gradient, mergeToCoarseGradT (CPU.hpp:946-959), cost (CPU.hpp:988), addLayerTGrad (CPU.hpp:816-894): all within wave 0 ----
        if (piece) gT[kk] = gTl + ((KN(KV, 0, kk) + KN(KV, 1, kk)) + KN(KV, 2, kk)) + dp.rho;     // + rho: CPU.hpp:989
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double sumTc = 0.0;
        if (kk < cN) {
            const int packed = (int)gCo[kk], iv = packed & 1023, fb = packed >> 10;   // (parked above: see the note at the host read)
            double sg = 0.0, tt = 0.0;
            for (int a = 0; a < iv; a++) { sg += gT[fb + a]; tt += Tf[fb + a]; }
            gCo[kk] = sg / iv;
            sumTc = tt;
        }
        const double wc = wave_sum_dpp(costAcc), wtt = wave_sum_dpp(sumTc);
        const double fval = wc + dp.rho * wtt;
        if (kk == 0) { if (!ro) f[b] = fval; red[0] = fval; }             // resident caller: the value travels through the mailbox, nothing to drain
        if (dp.soft) {
            if (kk < cN) {
                const double gi = gCo[kk] * dtt;
                if (gs) gs[kk] = gi; else g[x0 + kk] = gi;
                if (gpub && !defer_time_gpub) stg<SH>(gpub + kk, gi, gwt);
                if (tapped) { t_dg += gi * d_tau; t_xx += x_tau * x_tau; t_gg += gi * gi; }
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (kk == 0) {
                const int Ms1 = cN - 1;
                const double gTail = dp.sumT * gCo[Ms1];
                double expTauSum = 0.0, gFreeDotExpTau = 0.0;
                for (int i = 0; i < Ms1; i++) {
                    const double e = tau_to_T(xs[i], dp.c2 != 0);
                    expTauSum += e;
                    gFreeDotExpTau += e * (dp.sumT * gCo[i]);
                }
                const double den = expTauSum + 1.0;
                for (int i = 0; i < Ms1; i++) {
                    const double de = dT_dtau(xs[i], dp.c2 != 0);
                    const double gi = (dp.sumT * gCo[i] - gTail) * de / den - (gFreeDotExpTau - gTail * expTauSum) * de / (den * den);
                    if (gs) gs[i] = gi; else g[x0 + i] = gi;
                    if (gpub && !defer_time_gpub) stg<SH>(gpub + i, gi, gwt);
                    if (tapped) { t_dg += gi * dsv[i]; t_xx += xs[i] * xs[i]; t_gg += gi * gi; }
                }
            }
        }
    } else {
        // ---- addPropCtoP + addLayerPGrad (CPU.hpp:154-161, 897-928): waypoint w (= knot w+1) on a PAIR of lanes of the axis waves ----
        // with r_a = sc xi_a, sc = 2 / (1 + |xi|^2):  d f / d xi_a = xi_a (2 sc^2 (V_a . g) - 4 gdq / (1 + |xi|^2)^2),
        // gdq = 2 sc sum_a (V_a . g) xi_a^2  -  |xi|^2 and the weighted sum come out of ONE pass over the vertices, which the forward map of this
        // evaluation made already (cached: in LDS for the resident caller, in dp.wq_glob between the stage kernels)
        const int w = wpt;
        const bool wact = wact0;
        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
        if (wact) { g0 = KN(KP, 0, w + 1); g1 = KN(KP, 1, w + 1); g2 = KN(KP, 2, w + 1); }
        if (cached0) {
            const double s2 = wq1 * g0 + wq2 * g1 + wq3 * g2;
            FRX_STAMP_AX(30);
            const double gdq = w_c2sc * s2, kq = 4.0 * gdq * w_iq2, sc22 = w_sc22;
            FRX_STAMP_AX(31);
            if (wact) {
#pragma unroll
                for (int j = 0; j < WPF; j++)                              // the vertices fetched in front of the poll
                    if (sub + 2 * j < nv1w) {
                        const double dgv = pvx[j] * g0 + pvy[j] * g1 + pvz[j] * g2;
                        const double gi = pxv[j] * (sc22 * dgv - kq);
                        if (gs) gs[xbw - x0 + sub + 2 * j] = gi; else g[xbw + sub + 2 * j] = gi;
                        if (gpub) stg<SH>(gpub + (xbw - x0 + sub + 2 * j), gi, gwt);
                        t_dg += gi * pdd[j]; t_gg += gi * gi;
                    }
                for (int a0 = sub + 2 * WPF; a0 < nv1w; a0 += 8) {         // polytopes with more than 16 vertices: the rest as before
                    double xv[4], dgv[4], dd[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int a = min(a0 + 2 * j, nv1w - 1);
                        xv[j] = xiw[a]; dgv[j] = Vw[3 * (a + 1)] * g0 + Vw[3 * (a + 1) + 1] * g1 + Vw[3 * (a + 1) + 2] * g2;
                        dd[j] = tapped ? dsv[xbw - x0 + a] : 0.0;
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (a0 + 2 * j < nv1w) {
                            const double gi = xv[j] * (sc22 * dgv[j] - kq);
                            if (gs) gs[xbw - x0 + a0 + 2 * j] = gi; else g[xbw + a0 + 2 * j] = gi;
                            if (gpub) stg<SH>(gpub + (xbw - x0 + a0 + 2 * j), gi, gwt);
                            t_dg += gi * dd[j]; t_xx += xv[j] * xv[j]; t_gg += gi * gi;
                        }
                }
            }
        } else {
            // (no sums left behind by a forward map - a caller without dp.wq_glob: the two-pass form of round 2)
            const double *V = Vw, *xi = xiw;
            const int nv1 = nv1w, xb = xbw;
            double qn = 0.0, s2 = 0.0;
            if (wact)
                for (int a0 = sub; a0 < nv1; a0 += 8) {                   // four vertices per trip, their LDS reads in flight together (clamped, not predicated)
                    double xv[4], dgv[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { const int a = min(a0 + 2 * j, nv1 - 1); xv[j] = xi[a]; dgv[j] = V[3 * (a + 1)] * g0 + V[3 * (a + 1) + 1] * g1 + V[3 * (a + 1) + 2] * g2; }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (a0 + 2 * j < nv1) { const double x2 = xv[j] * xv[j]; qn += x2; s2 += dgv[j] * x2; }
                }
            FRX_STAMP_AX(30);
            qn += dpp_mov<0xB1>(qn); s2 += dpp_mov<0xB1>(s2);   // pair sums
            const double qp1 = qn + 1.0, iq = 1.0 / qp1, sc = 2.0 * iq;
            const double gdq = 2.0 * sc * s2, kq = 4.0 * gdq * (iq * iq), sc22 = 2.0 * sc * sc;
            FRX_STAMP_AX(31);
            if (wact)
                for (int a0 = sub; a0 < nv1; a0 += 8) {
                    double xv[4], dgv[4], dd[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int a = min(a0 + 2 * j, nv1 - 1);
                        xv[j] = xi[a]; dgv[j] = V[3 * (a + 1)] * g0 + V[3 * (a + 1) + 1] * g1 + V[3 * (a + 1) + 2] * g2;
                        dd[j] = tapped ? dsv[xb - x0 + a] : 0.0;
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (a0 + 2 * j < nv1) {
                            const double gi = xv[j] * (sc22 * dgv[j] - kq);
                            if (gs) gs[xb - x0 + a0 + 2 * j] = gi; else g[xb + a0 + 2 * j] = gi;
                            if (gpub) stg<SH>(gpub + (xb - x0 + a0 + 2 * j), gi, gwt);
                            t_dg += gi * dd[j]; t_xx += xv[j] * xv[j]; t_gg += gi * gi;
                        }
                }
        }
    }
    FRX_STAMP_AX(29);
    FRX_STAMP(23);
    // ---- line-search tap: what lbfgs.hpp:830 (g.d) and :1296-1297 (|x|, |g|) need, reduced here instead of in a separate launch ----
    if (tap.d != nullptr) {                                           // uniform over the grid
        const double w0 = wave_sum_dpp(t_dg), w1 = wave_sum_dpp(t_xx), w2 = wave_sum_dpp(t_gg);
        const int nw = nthr >> 6, w = k >> 6;
        double *red3 = rowbuf;                                        // the row buffer is not used by this path
        if ((k & 63) == 0) { red3[w] = w0; red3[nw + w] = w1; red3[2 * nw + w] = w2; }
        __syncthreads();
        if (defer_time_gpub && wave == 1 && kk < (dp.soft ? cN : cN - 1)) stg<SH>(gpub + kk, gs[kk], gwt);   // (see defer_time_gpub)
        if (k == 0 && ((tap_flags & DV_EVAL) || tap.lds_out)) {
            const double fval = red[0];
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (int i = 0; i < nw; i++) { a0 += red3[i]; a1 += red3[nw + i]; a2 += red3[2 * nw + i]; }
            if (tap.lds_out) { tap.lds_out[0] = fval; tap.lds_out[1] = a0; tap.lds_out[2] = a1; tap.lds_out[3] = a2; if (tap.early_cmd) { tap.lds_out[7] = __longlong_as_double((long long)early_word); tap.lds_out[6] = __longlong_as_double((long long)early_step); } }
            else { DvResult *r = tap.res + b; r->f = fval; r->dg = a0; r->xx = a1; r->gg = a2; }
        }
        if (k == 0 && tap.arrive) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // system scope: the result above is visible before the count moves
            if (atomicAdd(tap.arrive, 1u) + 1u == (unsigned)gridDim.x * tap.round) *tap.flag = tap.round;
        }
    }
    FRX_STAMP(24);
}

// SH: out20 (and T, C) were written by workgroups of the same launch.
template <bool SH, int NR = 0, int RB = 0>
__device__ __forceinline__ void backward_knot_body(const DevProblem &dp, const double *__restrict__ x, const double *__restrict__ Tin,
                                const double *__restrict__ Cin, const double *__restrict__ out20, double *__restrict__ f,
                                double *__restrict__ g, int maxCN, int maxXb, int maxVb, int nrow_rt, const double *__restrict__ pcrw, int nsteps,
                                const LineSearchTap &tap, int b, double *sm, const double *ct_lds = nullptr, const ResidentOps *ro = nullptr) {
    const int nrow = NR > 0 ? NR : nrow_rt;
    if (NR == 64 || (NR == 0 && nrow == 64 && blockDim.x == 256)) {              // <= 64 pieces: one wave per axis
        if (SH) backward_knot_wsp64<SH>(dp, x, Tin, Cin, out20, f, g, maxCN, maxXb, maxVb, pcrw, nsteps, tap, b, sm, ct_lds, ro);       // resident caller: the order built around the poll
        else backward_knot_wsp64_stage<SH, RB>(dp, x, Tin, Cin, out20, f, g, maxCN, maxXb, maxVb, pcrw, nsteps, tap, b, sm, ct_lds, ro);
        return;
    }
    if (NR == 64) return;
    const int k = threadIdx.x, nthr = NR > 0 ? 256 : (int)blockDim.x;
    const int p0 = dp.poff[b], N = dp.poff[b + 1] - p0;
    const int c0 = dp.coff[b], cN = dp.coff[b + 1] - c0;
    const int x0 = dp.xoff[b];
    double *rowbuf = sm;
    double *KP = rowbuf + (size_t)36 * nrow;          // reused: knot states, then end-of-piece adjoints, then mu
    double *KV = KP + 3 * (nrow + 1);
    double *KA = KV + 3 * (nrow + 1);
    double *Tf = KA + 3 * (nrow + 1);
    double *gT = Tf + nrow;
    double *gCo = gT + nrow;
    double *red = gCo + maxCN;                         // [2 * nthr/64] cross-wave partials
    double *xs = red + 2 * (nthr >> 6) + 2;
    double *vs = xs + maxXb;
    double *pw = vs + maxVb;                            // [nrow][nsteps*8+5] multipliers saved by k_forward_knot
    double *dsv = pw + (size_t)(nsteps * 8 + 5) * nrow; // [maxXb] search direction (only with a line-search tap)
    const bool pw_resident = ro && nrow == 64 && nthr == 256;
    if (ro) { xs = ro->xs; vs = ro->vs; dsv = ro->dsv; if (pw_resident) pw = ro->pw; }
    const bool tapped = tap.d != nullptr;
    const int tap_flags = (tapped && tap.flags) ? tap.flags[b] : 0;   // consumed by thread 0 at the very end
    double t_dg = 0.0, t_xx = 0.0, t_gg = 0.0;          // g.d, x.x, g.g over the elements this thread writes
    FRX_STAMP(16);
    // all global reads up front (see k_forward_knot)
    double h = 1.0, c[18], cb[18], o0 = 0.0, o1 = 0.0, r_tl[3] = {0, 0, 0};
    int r_wnv = 1, r_wvb = 0, r_wxb = 0;
    const int cv0 = __builtin_amdgcn_readfirstlane(dp.cvoff[b]);
    if (k < N) {
        const double *ci = Cin + (size_t)(p0 + k) * 18;
        const double *o = out20 + (size_t)(p0 + k) * 20;
        o0 = ldg<SH>(o); o1 = ldg<SH>(o + 1);
#pragma unroll
        for (int q = 0; q < 18; q++) cb[q] = ldg<SH>(o + 2 + q);
        if (ct_lds) {                                           // this workgroup's own forward pass left them in LDS
            h = ct_lds[k * 19 + 18];
#pragma unroll
            for (int q = 0; q < 18; q++) c[q] = ct_lds[k * 19 + q];
        } else {
            h = ldg<SH>(Tin + p0 + k);
#pragma unroll
            for (int q = 0; q < 18; q++) c[q] = ldg<SH>(ci + q);
        }
    }
    if ((k >> 2) < N - 1) { const int gw = p0 - b + (k >> 2); r_wnv = dp.wp_nv[gw]; r_wvb = dp.wp_vbeg[gw]; r_wxb = dp.wp_xbeg[gw]; }   // quad k>>2 = waypoint
    if (k < 3) { r_tl[0] = dp.tailPVA[b * 9 + k]; r_tl[1] = dp.tailPVA[b * 9 + 3 + k]; r_tl[2] = dp.tailPVA[b * 9 + 6 + k]; }
    {
        const int nx = dp.xoff[b + 1] - x0;
        const int v0 = dp.cvoff[b], nvd = 3 * (dp.cvoff[b + 1] - v0);
        const double *vsrc = dp.vrec + 3 * (size_t)v0;
        if (!ro) {
            stage_to_lds<4>(xs, x + x0, nx, k, nthr);
            stage_to_lds<8>(vs, vsrc, nvd, k, nthr);
            if (tapped) stage_to_lds<4>(dsv, tap.d + x0, nx, k, nthr);
        }
    }
    if (!pw_resident) {   // saved multipliers of this candidate: one contiguous block, 16-byte loads by all threads (a single batch)
        const int ws = nsteps * 8 + 4, n2 = (N * ws) >> 1;                 // ws is even
        const double2 *src = (const double2 *)(pcrw + (size_t)p0 * ws);
        for (int i0 = k; i0 < n2; i0 += 8 * nthr) {                          // same batching as stage_to_lds, 16-byte loads
            double2 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i2 = i0 + u * nthr; tmp[u] = src[i2 < n2 ? i2 : n2 - 1]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i2 = i0 + u * nthr;
                if (i2 < n2) {
                    const int e = 2 * i2, kn = e / ws, ff = e - kn * ws;        // ws even: both halves belong to the same knot
                    pw[kn * (ws + 1) + ff] = tmp[u].x; pw[kn * (ws + 1) + ff + 1] = tmp[u].y;
                }
            }
        }
    }

    FRX_STAMP(17);
    // ---- piece-local: load, jerk energy + gradients (CPU.hpp:507-520, 65-95), cbar = d f / d c ----
    double gTl = 0.0, costAcc = 0.0;
    if (k < N) {
        Tf[k] = h;
        const double t1 = h, t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
        const double *c3 = c + 9, *c4 = c + 12, *c5 = c + 15;
        const double s33 = dot3(c3, c3), s43 = dot3(c4, c3), s44 = dot3(c4, c4), s53 = dot3(c5, c3), s54 = dot3(c5, c4), s55 = dot3(c5, c5);
        costAcc = o0 + (36.0 * s33 * t1 + 144.0 * s43 * t2 + 192.0 * s44 * t3 + 240.0 * s53 * t3 + 720.0 * s54 * t4 + 720.0 * s55 * t5);
        gTl = o1 + (36.0 * s33 + 288.0 * s43 * t1 + 576.0 * s44 * t2 + 720.0 * s53 * t2 + 2880.0 * s54 * t3 + 3600.0 * s55 * t4);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            cb[9 + d] += 72.0 * c3[d] * t1 + 144.0 * c4[d] * t2 + 240.0 * c5[d] * t3;
            cb[12 + d] += 144.0 * c3[d] * t2 + 384.0 * c4[d] * t3 + 720.0 * c5[d] * t4;
            cb[15 + d] += 240.0 * c3[d] * t3 + 720.0 * c4[d] * t4 + 1440.0 * c5[d] * t5;
        }
        // knot states recovered from the coefficients: p = c0, v = c1, a = 2 c2 at the piece start
#pragma unroll
        for (int ax = 0; ax < 3; ax++) { KN(KP, ax, k) = c[ax]; KN(KV, ax, k) = c[3 + ax]; KN(KA, ax, k) = 2.0 * c[6 + ax]; }
    }
    if (k < 3) { KN(KP, k, N) = r_tl[0]; KN(KV, k, N) = r_tl[1]; KN(KA, k, N) = r_tl[2]; }
    unsigned long long early_word = 0, early_step = 0;
    if (tap.early_cmd && k == 0) { early_word = __hip_atomic_load(tap.early_cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); early_step = __hip_atomic_load(tap.early_cmd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // behind the first loads (see LineSearchTap)
    __syncthreads();

    FRX_STAMP(18);
    // ---- Hermite adjoint per piece; end-of-piece parts go to the next knot through LDS ----
    double p1[3], v1[3], a1[3], sP[3] = {0, 0, 0}, sV[3] = {0, 0, 0}, sA[3] = {0, 0, 0}, eP[3], eV[3], eA[3], hb = 0.0;
    if (k < N) {
#pragma unroll
        for (int ax = 0; ax < 3; ax++) { p1[ax] = KN(KP, ax, k + 1); v1[ax] = KN(KV, ax, k + 1); a1[ax] = KN(KA, ax, k + 1); }
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
            double cba[6], db[6], hba;
#pragma unroll
            for (int q = 0; q < 6; q++) cba[q] = cb[q * 3 + ax];
            hermite_adjoint(h, c[ax], c[3 + ax], 2.0 * c[6 + ax], p1[ax], v1[ax], a1[ax], cba, db, hba);
            sP[ax] = db[0]; sV[ax] = db[1]; sA[ax] = db[2];
            eP[ax] = db[3]; eV[ax] = db[4]; eA[ax] = db[5];
            hb += hba;
        }
    }
    __syncthreads();                                    // everyone has read the knot states
    if (k < N) {
#pragma unroll
        for (int ax = 0; ax < 3; ax++) { KN(KP, ax, k + 1) = eP[ax]; KN(KV, ax, k + 1) = eV[ax]; KN(KA, ax, k + 1) = eA[ax]; }
    }
    __syncthreads();

    FRX_STAMP(19);
    // ---- mu = K^-1 wbar: same SPD knot matrix as the forward pass, whose reduction multipliers are re-used ----
    double muv[3] = {0, 0, 0}, mua[3] = {0, 0, 0}, pbk[3] = {0, 0, 0};
    {
        double rr[6] = {0, 0, 0, 0, 0, 0};
        if (k >= 1 && k <= N - 1) {
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                rr[ax] = sV[ax] + KN(KV, ax, k);
                rr[3 + ax] = sA[ax] + KN(KA, ax, k);
                pbk[ax] = sP[ax] + KN(KP, ax, k);          // direct d f / d p_k (both adjacent pieces)
            }
        }
    FRX_STAMP(20);
        const bool wsp = (nrow == 64 || nrow == 128) && nthr == 256; // one wave per axis (pcr_waves_wg / pcr_waves2_wg) writes mu to KV / KA itself
        if (wsp) {
            if (k >= 1 && k <= N - 1) {
#pragma unroll
                for (int i = 0; i < 6; i++) rowbuf[(size_t)24 * nrow + i * nrow + k] = rr[i];
            }
            __syncthreads();
            if (nrow == 64) pcr_waves_wg(rowbuf, nrow, k, N, pw, nsteps * 8 + 5, nsteps, KV, KA);
            else pcr_waves2_wg(rowbuf, nrow, k, N, false, pw, nsteps * 8 + 5, nullptr, 0, 0, nsteps, KV, KA);
        } else {
            pcr_apply_wg(rowbuf, nrow, k, N, rr, pw, nsteps);
        }
    FRX_STAMP(21);
#pragma unroll
        for (int ax = 0; ax < 3; ax++) { muv[ax] = rr[ax]; mua[ax] = rr[3 + ax]; }
        // publish mu (zero at the fixed end knots) — reuse KV/KA
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
            if (wsp) { if (k == 0 || k == N) { KN(KV, ax, k) = 0.0; KN(KA, ax, k) = 0.0; } }
            else if (k <= N) { KN(KV, ax, k) = (k >= 1 && k <= N - 1) ? muv[ax] : 0.0; KN(KA, ax, k) = (k >= 1 && k <= N - 1) ? mua[ax] : 0.0; }
        }
    }
    if (k == 0 && N == nrow) {
#pragma unroll
        for (int ax = 0; ax < 3; ax++) { KN(KV, ax, N) = 0.0; KN(KA, ax, N) = 0.0; }
    }
    __syncthreads();
    // ---- through the knot system: duration term and d f / d(p_{k+1} - p_k) ----
    double dlb[3] = {0, 0, 0};
    if (k < N) {
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
            dlb[ax] = knot_adjoint_piece(h, p1[ax] - c[ax], c[3 + ax], 2.0 * c[6 + ax], v1[ax], a1[ax], KN(KV, ax, k), KN(KA, ax, k),
                                         KN(KV, ax, k + 1), KN(KA, ax, k + 1), hb);
        gT[k] = gTl + hb + dp.rho;                     // + rho: CPU.hpp:989
    }
    __syncthreads();
    if (k < N) {
#pragma unroll
        for (int ax = 0; ax < 3; ax++) KN(KP, ax, k + 1) = dlb[ax];       // +dl to knot k+1
    }
    __syncthreads();
    double gq[3] = {0, 0, 0};
    if (k >= 1 && k <= N - 1) {
#pragma unroll
        for (int ax = 0; ax < 3; ax++) gq[ax] = pbk[ax] + KN(KP, ax, k) - dlb[ax];   // -dl of the piece starting here
    }

    FRX_STAMP(22);
    // ---- cost (CPU.hpp:988) and mergeToCoarseGradT (CPU.hpp:946-959) ----
    double sumTc = 0.0, fval = 0.0;
    for (int i = k; i < cN; i += nthr) {
        const int gc = c0 + i;
        const int iv = dp.coarse_iv[gc], fb = dp.coarse_fbeg[gc] - p0;
        double s = 0.0, tt = 0.0;
        for (int a = 0; a < iv; a++) { s += gT[fb + a]; tt += Tf[fb + a]; }
        gCo[i] = s / iv;
        sumTc += tt;
    }
    {
        const double wc = wave_sum_dpp(costAcc), wt = wave_sum_dpp(sumTc);
        const int nw = nthr >> 6, w = k >> 6;
        if ((k & 63) == 0) { red[w] = wc; red[nw + w] = wt; }
        __syncthreads();
        if (k == 0) {
            double tc = 0.0, tt = 0.0;
            for (int i = 0; i < nw; i++) { tc += red[i]; tt += red[nw + i]; }
            fval = tc + dp.rho * tt;
            f[b] = fval;
        }
    }
    FRX_STAMP(23);
    // ---- addLayerTGrad (CPU.hpp:816-894) ----
    if (dp.soft) {
        for (int i = k; i < cN; i += nthr) {
            const double gi = gCo[i] * dT_dtau(xs[i], dp.c2 != 0);
            g[x0 + i] = gi; if (ro && ro->gs) ro->gs[i] = gi;
            if (ro && ro->gpub) stg<SH>(ro->gpub + i, gi, ro->gwt);
            if (tapped) { t_dg += gi * dsv[i]; t_xx += xs[i] * xs[i]; t_gg += gi * gi; }
        }
    } else if (k == 0) {
        const int Ms1 = cN - 1;
        const double gTail = dp.sumT * gCo[Ms1];
        double expTauSum = 0.0, gFreeDotExpTau = 0.0;
        for (int i = 0; i < Ms1; i++) {
            const double e = tau_to_T(xs[i], dp.c2 != 0);
            expTauSum += e;
            gFreeDotExpTau += e * (dp.sumT * gCo[i]);
        }
        const double den = expTauSum + 1.0;
        for (int i = 0; i < Ms1; i++) {
            const double de = dT_dtau(xs[i], dp.c2 != 0);
            const double gi = (dp.sumT * gCo[i] - gTail) * de / den - (gFreeDotExpTau - gTail * expTauSum) * de / (den * den);
            g[x0 + i] = gi; if (ro && ro->gs) ro->gs[i] = gi;
            if (ro && ro->gpub) stg<SH>(ro->gpub + i, gi, ro->gwt);
            if (tapped) { t_dg += gi * dsv[i]; t_xx += xs[i] * xs[i]; t_gg += gi * gi; }
        }
    }
    // ---- addPropCtoP + addLayerPGrad (CPU.hpp:154-161, 897-928): waypoint w (= knot w+1) on a quad of lanes ----
    __syncthreads();
    if (k >= 1 && k <= N - 1) {
#pragma unroll
        for (int ax = 0; ax < 3; ax++) KN(KP, ax, k) = gq[ax];      // d f / d q_k to the quad that owns the waypoint
    }
    __syncthreads();
    for (int w0 = 0; w0 < N - 1; w0 += nthr / 4) {
        const int w = w0 + (k >> 2), sub = k & 3;
        const bool wact = w < N - 1;
        const double *V = vs, *xi = xs;
        int nv1 = 0, xb = 0;
        double g0 = 0.0, g1 = 0.0, g2 = 0.0, qn = 0.0, gdq = 0.0;
        if (wact) {
            int wnv = r_wnv, wvb = r_wvb, wxb = r_wxb;
            if (w0 > 0) { const int gw = p0 - b + w; wnv = dp.wp_nv[gw]; wvb = dp.wp_vbeg[gw]; wxb = dp.wp_xbeg[gw]; }
            nv1 = wnv - 1; xb = wxb;
            V = vs + 3 * (wvb - cv0) + (ro ? ro->vskew * w : 0);
            xi = xs + (xb - x0);
            g0 = KN(KP, 0, w + 1); g1 = KN(KP, 1, w + 1); g2 = KN(KP, 2, w + 1);
            for (int a = sub; a < nv1; a += 4) qn += xi[a] * xi[a];
        }
        qn = quad_sum(qn);
        const double qp1 = qn + 1.0, qp1sq = qp1 * qp1, sc = 2.0 / qp1;
        if (wact)
            for (int a = sub; a < nv1; a += 4) {
                const double gdr = (V[3 * (a + 1)] * g0 + V[3 * (a + 1) + 1] * g1 + V[3 * (a + 1) + 2] * g2) * (sc * xi[a]) * 2.0;
                gdq += gdr * xi[a];
            }
        gdq = quad_sum(gdq);
        if (wact)
            for (int a = sub; a < nv1; a += 4) {
                const double gdr = (V[3 * (a + 1)] * g0 + V[3 * (a + 1) + 1] * g1 + V[3 * (a + 1) + 2] * g2) * (sc * xi[a]) * 2.0;
                const double gi = gdr * 2.0 / qp1 - xi[a] * 4.0 * gdq / qp1sq;
                g[xb + a] = gi; if (ro && ro->gs) ro->gs[xb - x0 + a] = gi;
                if (ro && ro->gpub) stg<SH>(ro->gpub + (xb - x0 + a), gi, ro->gwt);
                if (tapped) { t_dg += gi * dsv[xb - x0 + a]; t_xx += xi[a] * xi[a]; t_gg += gi * gi; }
            }
    }
    // ---- line-search tap: what lbfgs.hpp:830 (g.d) and :1296-1297 (|x|, |g|) need, reduced here instead of in a separate launch ----
    if (tap.d != nullptr) {                                           // uniform over the grid
        const double w0 = wave_sum_dpp(t_dg), w1 = wave_sum_dpp(t_xx), w2 = wave_sum_dpp(t_gg);
        const int nw = nthr >> 6, w = k >> 6;
        __syncthreads();                                              // red[] was last read for f[b]
        double *red3 = rowbuf;                                        // row buffer is dead by now
        if ((k & 63) == 0) { red3[w] = w0; red3[nw + w] = w1; red3[2 * nw + w] = w2; }
        __syncthreads();
        if (k == 0 && ((tap_flags & DV_EVAL) || tap.lds_out)) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (int i = 0; i < nw; i++) { a0 += red3[i]; a1 += red3[nw + i]; a2 += red3[2 * nw + i]; }
            if (tap.lds_out) { tap.lds_out[0] = fval; tap.lds_out[1] = a0; tap.lds_out[2] = a1; tap.lds_out[3] = a2; if (tap.early_cmd) { tap.lds_out[7] = __longlong_as_double((long long)early_word); tap.lds_out[6] = __longlong_as_double((long long)early_step); } }
            else { DvResult *r = tap.res + b; r->f = fval; r->dg = a0; r->xx = a1; r->gg = a2; }
        }
        if (k == 0 && tap.arrive) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // system scope: the result above is visible before the count moves
            if (atomicAdd(tap.arrive, 1u) + 1u == (unsigned)gridDim.x * tap.round) *tap.flag = tap.round;
        }
    }
    FRX_STAMP(24);
#undef KN
}
FRX_KERNEL_LINKAGE __global__ __launch_bounds__(256) void k_backward_knot(DevProblem dp, const double *__restrict__ x, const double *__restrict__ Tin,
                                const double *__restrict__ Cin, const double *__restrict__ out20, double *__restrict__ f,
                                double *__restrict__ g, int maxCN, int maxXb, int maxVb, int nrow, const double *__restrict__ pcrw, int nsteps,
                                LineSearchTap tap) {
    extern __shared__ double sm[];
    if (dp.cand_active && !(dp.cand_active[blockIdx.x] & DV_EVAL)) {           // skipped candidate: only the arrival count
        if (threadIdx.x == 0 && tap.arrive && atomicAdd(tap.arrive, 1u) + 1u == (unsigned)gridDim.x * tap.round) *tap.flag = tap.round;
        return;
    }
    backward_knot_body<false>(dp, x, Tin, Cin, out20, f, g, maxCN, maxXb, maxVb, nrow, pcrw, nsteps, tap, blockIdx.x, sm);
}
FRX_KERNEL_LINKAGE __global__ __launch_bounds__(256) void k_backward_knot64(DevProblem dp, const double *__restrict__ x, const double *__restrict__ Tin,
                                const double *__restrict__ Cin, const double *__restrict__ out20, double *__restrict__ f,
                                double *__restrict__ g, int maxCN, int maxXb, int maxVb, const double *__restrict__ pcrw, int nsteps,
                                LineSearchTap tap) {
    extern __shared__ double sm[];
    if (dp.cand_active && !(dp.cand_active[blockIdx.x] & DV_EVAL)) {           // skipped candidate: only the arrival count
        if (threadIdx.x == 0 && tap.arrive && atomicAdd(tap.arrive, 1u) + 1u == (unsigned)gridDim.x * tap.round) *tap.flag = tap.round;
        return;
    }
    backward_knot_body<false, 64>(dp, x, Tin, Cin, out20, f, g, maxCN, maxXb, maxVb, 64, pcrw, nsteps, tap, blockIdx.x, sm);
}
#undef ROW2

#undef BAND
} // namespace frx
