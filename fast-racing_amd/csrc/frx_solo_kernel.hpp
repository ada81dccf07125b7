// ONE LAUNCH PER EVALUATION for batches LARGER than the chip holds as clusters (frx_objective_eval[_device] and the per-stage rounds of frx_optimize in the window of
// batch sizes where it measures faster; the reference's objectiveFunc, se3gcopter_cpu.hpp:961-1000): one WORKGROUP per candidate runs the forward map, the penalty
// integral of its own pieces and the adjoint back to back - the three stage bodies of frx_kernels.hpp - where the stage path launches three grids.
//
// Why (profiles/r06_knot_sweep.jsonl): at Monte-Carlo scale (512 candidates per GPU) an evaluation was 11.7 + 14.8 + 11.5 us.  The knot
// kernels are one dependent chain per candidate (7.6 us with 32 candidates) that two co-resident workgroups per CU and the dispatch of 512 workgroups stretch to
// 11.7; every launch pays its own ramp (~2.4 us to dispatch 512 workgroups of 57 KB LDS), its own drain and a trip of its stage buffers through memory:
// 31.6 + 22.8 + 39.3 MB of counted traffic for 35.8 MB of algorithmic bytes, two thirds of it (C, T), out20, the saved reduction multipliers and the re-staged
// variables and polytopes on their way from one launch to the next.
//
// Form 2 of this kernel (round 6, late): NOTHING of that travels any more but the penalty partials.  The workgroup's LDS holds, for the whole launch, the (C, T) copy, x,
// the waypoint polytopes, the reduction multipliers and the waypoint sums (ResidentOps, as the one-launch cluster's leader keeps them) - the forward map fills them, the
// penalty phase reads (C, T) from there, the adjoint finds everything but the partials in place.  What made that fit TWO workgroups on a CU (80 KB each; form 1 kept
// nothing across a phase boundary because the straightforward layout needs 84.9 KB): the bodies' row buffer cut from 36 x 64 to the 1 540 doubles the <= 64-piece path
// uses (template parameter RB), the search direction of a round's line-search tap parked in that buffer (the adjoint does not use it), and a penalty phase that lives
// in the bodies' scratch and the polytopes' place (34 KB) - corridor blocks of ONE pass at a time (requested a pass ahead, held in registers), the 20 partials through the
// LDS transpose in two halves of ten as in k_penalty_lat2 (four quarters of five inside the scratch alone measured 11 k cycles slower per candidate: 40 barriers and four
// dependent sums per pass) - after which the polytopes (constants of the handle) are staged once more for the adjoint's waypoint layer.  Same samples, same fixed-order sums, same bodies: f, gradient and plans stay BIT-IDENTICAL to the three stage launches (tests/test_gpu_solo.py).
#pragma once
#include "frx_kernels.hpp"

namespace frx {

struct SoloArgs {
    const double *x; double *T, *C, *out20, *f, *g;
    double *pcrw;
    int maxCN, maxXb, maxVb, nsteps, lpp, ppg, Kmax, maxN;
};

} // namespace frx
#include "frx_solo_layout.hpp"
namespace frx {

// corridor blocks of the pieces [q0, q0 + np) of the candidate whose first piece is p0: NT 16-byte trips per thread into registers (clamped, unconditional)
template <int NT>
__device__ __forceinline__ void solo_fetch_corridor(const DevProblem &dp, int p0, int q0, int np, int hstride, double2 (&hv)[NT]) {
    const double2 *h2 = (const double2 *)(dp.hblk + (size_t)(p0 + q0) * hstride);
    const int nh2 = (np * hstride) >> 1;
#pragma unroll
    for (int u = 0; u < NT; u++) { const int i = (int)threadIdx.x + u * 256; hv[u] = h2[nh2 > 0 ? (i < nh2 ? i : nh2 - 1) : 0]; }
}

template <int LPP>
__device__ __forceinline__ void solo_penalty_phase(const DevProblem &dp, const double *ctl, double *out20, int lpp_rt, int ppg, int Kmax, int p0, int N, double *scratch, double2 (&hv)[3]) {
    constexpr int nthr = 256, NT = 3;                                           // (three 16-byte trips per thread cover a pass of 15 pieces up to K = 24; beyond: the tail loop)
    const int lpp = LPP ? LPP : lpp_rt;
    const int lane = threadIdx.x;
    const int hstride = (Kmax + 1) * 4;
    const int ppass = ppg < N ? ppg : N;
    double *hS = scratch, *red = scratch + (size_t)ppass * hstride;
    const int pl = lane / lpp, jl = lane - pl * lpp;
    double *mine = red + lane * SOLO_QS;
#pragma unroll 1
    for (int q0 = 0; q0 < N; q0 += ppg) {
        const int npieces = min(ppg, N - q0);
        const bool exists = pl < npieces;
        const int pfl = (dp.piece_active && exists) ? dp.piece_active[p0 + q0 + pl] : DV_EVAL;
        const bool active = exists && (pfl & DV_EVAL);
        __syncthreads();                                                        // the previous pass's second half has been summed (first pass: the forward map is done with this scratch)
        {   // this pass's corridor blocks: from the registers they were requested into a pass ago (the first ones at the kernel's entry)
            const int nh2 = (npieces * hstride) >> 1;
#pragma unroll
            for (int u = 0; u < NT; u++) { const int i = lane + u * nthr; if (i < nh2) { hS[2 * i] = hv[u].x; hS[2 * i + 1] = hv[u].y; } }
            const double2 *h2 = (const double2 *)(dp.hblk + (size_t)(p0 + q0) * hstride);
#pragma unroll 1
            for (int i = lane + NT * nthr; i < nh2; i += nthr) { const double2 v = h2[i]; hS[2 * i] = v.x; hS[2 * i + 1] = v.y; }
        }
        if (q0 + ppg < N) solo_fetch_corridor<NT>(dp, p0, q0 + ppg, min(ppg, N - q0 - ppg), hstride, hv);   // the next pass's, under this pass's samples
        __syncthreads();
        double o[20];
        if (active) {
            LdsView c(ctl + (q0 + pl) * 19), hb(hS + (size_t)pl * hstride);
            const int K = (int)hb[3], kappa = dp.kappa;
            const double step = ctl[(q0 + pl) * 19 + 18] / kappa;             // the piece's STEP, CPU.hpp:245 (the stage kernels divide the same two numbers)
            penalty_sample_partials<true>(dp, c, hb, K, Kmax, step * jl, step, (jl == 0 || jl == kappa) ? 0.5 : 1.0, dp.inv_kappa, jl, o);
        }
#pragma unroll
        for (int qt = 0; qt < 20 / SOLO_QV; qt++) {
            if (qt) __syncthreads();                                            // the half before has been summed: its slots are free
            if (active) {
#pragma unroll
                for (int i = 0; i < SOLO_QV; i++) mine[i] = o[SOLO_QV * qt + i];
            } else if (exists) {                                                // a piece that is switched off this round: its partials are zeros
#pragma unroll
                for (int i = 0; i < SOLO_QV; i++) mine[i] = 0.0;
            }
            __syncthreads();
#pragma unroll 1
            for (int idx = lane; idx < npieces * SOLO_QV; idx += nthr) {
                const int p2 = idx / SOLO_QV, v = idx - p2 * SOLO_QV;
                const double *src = red + (p2 * lpp) * SOLO_QS + v;
                double s = 0.0;                                                 // one dependent chain of additions in sample order, like penalty_reduce; the reads sixteen at a time
                int l = 0;
                for (; l + 16 <= lpp; l += 16) {
                    double bb[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) bb[j] = src[(l + j) * SOLO_QS];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 16; j++) s += bb[j];
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll 4
                for (; l < lpp; l++) s += src[l * SOLO_QS];
                out20[(size_t)(p0 + q0 + p2) * 20 + SOLO_QV * qt + v] = s;
            }
        }
    }
}

template <int LPP>
__global__ __launch_bounds__(256, 2) void k_eval_solo(DevProblem dp, SoloArgs a, LineSearchTap tap) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int b = blockIdx.x, t = threadIdx.x;
    if (dp.cand_active && !(dp.cand_active[b] & DV_EVAL)) {                    // skipped candidate: only the arrival count (k_backward_knot64)
        if (t == 0 && tap.arrive && atomicAdd(tap.arrive, 1u) + 1u == (unsigned)gridDim.x * tap.round) *tap.flag = tap.round;
        return;
    }
    // (cycle stamps of candidate 0, frx_profile_phases with the form forced: slots 7 / 13 / 14 / 15 = entry, forward map done, penalty phase done, end - the bodies' own stamps use 0-6, 8-12 and 16-31)
#define SOLO_STAMP(slot) do { if (dp.stamps && b == 0 && t == 0) dp.stamps[slot] = (long long)__builtin_readcyclecounter(); } while (0)
    const SoloLds L = solo_lds(a.maxN, a.maxXb, a.maxVb, a.maxCN, a.nsteps, a.ppg, a.Kmax);
    double *ctl = sm + L.ctl, *ev = sm + L.ev;
    const int p0 = dp.poff[b], N = dp.poff[b + 1] - p0;
    const int hstride = (a.Kmax + 1) * 4;
    double2 hv[3];
    solo_fetch_corridor<3>(dp, p0, 0, min(a.ppg, N), hstride, hv);            // the first pass's corridor blocks wait in registers through the forward map (constants of the handle)
    SOLO_STAMP(7);
    ResidentOps ro;
    ro.xs = sm + L.xs; ro.vs = sm + L.vs; ro.dsv = ev + 16; ro.pw = sm + L.pw; ro.gs = nullptr; ro.vskew = 0; ro.wq = sm + L.wq; ro.gpub = nullptr; ro.gwt = true;
    // forward map: stages x and the polytopes into the resident arrays itself (MODE 1), keeps the multipliers and the waypoint sums there, collects (C, T) in ctl and
    // sends nothing of them to global memory (MODE 4)
    forward_knot_body<false, 64, 5, SOLO_RB>(dp, a.x, a.T, a.C, a.maxCN, a.maxXb, a.maxVb, 64, nullptr, a.nsteps, b, ev, ctl, true, &ro);
    SOLO_STAMP(13);
    solo_penalty_phase<LPP>(dp, ctl, a.out20, a.lpp, a.ppg, a.Kmax, p0, N, sm + L.vs, hv);
    __syncthreads();                                                            // out20 of this candidate is out (this workgroup's own stores, read back by the adjoint); the scratch is free
    SOLO_STAMP(14);
    {   // the polytopes again (the transpose square lay over them), and - in a round of frx_optimize - the search direction for the tap's g.d, parked in the row buffer the
        // adjoint does not use.  (Requested under the last penalty pass and held in registers, with the corridor blocks double-buffered: 198 instead of 171 VGPRs and a
        // penalty phase 2.8 k cycles longer - measured, not kept.)
        const int v0 = dp.cvoff[b], nvd = 3 * (dp.cvoff[b + 1] - v0);
        stage_to_lds<8>(ro.vs, dp.vrec + 3 * (size_t)v0, nvd, t, 256);
        if (tap.d != nullptr) { const int x0 = dp.xoff[b], nx = dp.xoff[b + 1] - x0; stage_to_lds<4>(ro.dsv, tap.d + x0, nx, t, 256); }
        __syncthreads();
    }
    backward_knot_body<false, 64, SOLO_RB>(dp, a.x, a.T, a.C, a.out20, a.f, a.g, a.maxCN, a.maxXb, a.maxVb, 64, nullptr, a.nsteps, tap, b, ev, ctl, &ro);
    __syncthreads();
    if (t == 0) a.f[b] = ev[SOLO_RB + 9 * 65 + 2 * 64 + a.maxCN];             // `red[0]` of backward_knot_wsp64_stage: with resident operands the body leaves f to its caller
    SOLO_STAMP(15);
#undef SOLO_STAMP
}

} // namespace frx
