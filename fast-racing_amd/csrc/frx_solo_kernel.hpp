// ONE LAUNCH PER EVALUATION for batches LARGER than the chip holds as clusters (frx_objective_eval[_device] and the per-stage rounds of frx_optimize from
// `solo_min_B` candidates on; the reference's objectiveFunc, se3gcopter_cpu.hpp:961-1000): one WORKGROUP per candidate runs the forward map, the penalty
// integral of its own pieces and the adjoint back to back - the three stage bodies of frx_kernels.hpp, unchanged - where the stage path launches three grids.
//
// Why (profiles/r06_knot_sweep.jsonl): at Monte-Carlo scale (512 candidates per GPU) an evaluation was 11.7 + 14.8 + 11.5 us.  The knot
// kernels are one dependent chain per candidate (7.6 us with 32 candidates) that two co-resident workgroups per CU and the dispatch of 512 workgroups stretch to
// 11.7; every launch pays its own ramp (~2.4 us to dispatch 512 workgroups of 57 KB LDS), its own drain and, because the L2s of the eight XCDs are only coherent
// through memory, a write-back of its stage buffers at its end and a fetch at the start of the next: 31.6 + 22.8 + 39.3 MB of counted traffic for 35.8 MB of
// algorithmic bytes, two thirds of it (C, T), out20 and the saved reduction multipliers on their way from one launch to the next.  Here a candidate's stage
// buffers are written and read back by the SAME workgroup: one ramp, one drain, and the integrator's samples of one workgroup run under the latency-bound knot
// phases of its neighbour on the CU.  (Counted traffic of the solo launch at 512 candidates: 81.0 MB against 94.0 MB for the three launches - the stage buffers
// still travel, a store does not leave its line behind in the L2 for the load that follows; keeping them in LDS does not fit two workgroups on a CU)
//
// Penalty phase: the candidate's coefficients, steps and corridor blocks are staged ONCE (one memory trip for all its pieces), then ceil(N / ppg) passes of
// ppg = floor(256 / (kappa + 1)) pieces each - lane = one quadrature sample, one LDS transpose and the fixed-order sums of the stage kernels (penalty_reduce): the
// partials are BIT-IDENTICAL to the stage kernel's, and so are f and the gradient (same bodies on the same inputs).
// LDS: max(forward map, adjoint, penalty phase) - the phases overlay each other; nothing is kept in LDS across a phase boundary in this first form.
#pragma once
#include "frx_kernels.hpp"

namespace frx {

struct SoloArgs {
    const double *x; double *T, *C, *out20, *f, *g;
    double *pcrw;
    int maxCN, maxXb, maxVb, nsteps, lpp, ppg, Kmax;
    int dbg;                                 // measurements only (FRX_SOLO_DEBUG): bit 0 = the forward map does not save its multipliers (the adjoint reads those of an earlier evaluation at the same point)
};

// doubles of dynamic LDS the penalty phase needs: cS[maxN*18] | tS[maxN] | hS[maxN*(Kmax+1)*4] | red[256*21]
__host__ __device__ inline size_t solo_pen_lds(int maxN, int Kmax) { return (size_t)maxN * 19 + (size_t)maxN * (Kmax + 1) * 4 + (size_t)256 * 21 + 2; }

template <int LPP>
__device__ __forceinline__ void solo_penalty_phase(const DevProblem &dp, const double *T, const double *C, double *out20, int lpp_rt, int ppg, int Kmax, int b, double *sm, const double2 (&hv)[5]) {
    constexpr int nthr = 256;
    const int lpp = LPP ? LPP : lpp_rt;
    const int lane = threadIdx.x;
    const int p0 = dp.poff[b], N = dp.poff[b + 1] - p0;
    const int hstride = (Kmax + 1) * 4;
    double *cS = sm, *tS = cS + N * 18, *hS = tS + N + (N & 1), *red = hS + (size_t)N * hstride;
    {   // the whole candidate in one batch of 16-byte loads (N <= 64: at most 3 + 1 + 5 trips per thread at K = 8)
        const double2 *h2 = (const double2 *)(dp.hblk + (size_t)p0 * hstride), *c2 = (const double2 *)(C + (size_t)p0 * 18);
        const int nh2 = (N * hstride) >> 1, nc2 = (N * 18) >> 1;
        double2 cv[3];                                                          // (hv: the first five trips of the corridor blocks, requested at the kernel's entry - solo_prefetch_corridor)
        double tv = 0.0;
#pragma unroll
        for (int u = 0; u < 3; u++) { const int i = lane + u * nthr; cv[u] = c2[i < nc2 ? i : nc2 - 1]; }
        if (lane < N) tv = T[p0 + lane];
#pragma unroll
        for (int u = 0; u < 3; u++) { const int i = lane + u * nthr; if (i < nc2) { cS[2 * i] = cv[u].x; cS[2 * i + 1] = cv[u].y; } }
#pragma unroll
        for (int u = 0; u < 5; u++) { const int i = lane + u * nthr; if (i < nh2) { hS[2 * i] = hv[u].x; hS[2 * i + 1] = hv[u].y; } }
        if (lane < N) tS[lane] = tv / dp.kappa;                                 // the piece's STEP, CPU.hpp:245
#pragma unroll 1
        for (int i = lane + 5 * nthr; i < nh2; i += nthr) { const double2 v = h2[i]; hS[2 * i] = v.x; hS[2 * i + 1] = v.y; }      // (corridor blocks of more than 8 half-spaces)
    }
    const int pl = lane / lpp, jl = lane - pl * lpp;
    // ONE-phase transpose here (the stage kernel's two halves exist to fit four workgroups of 29 KB on a CU; this workgroup owns 77 KB for the forward map and the
    // adjoint anyway): all 20 partials of a lane in one [256][21] square, two barriers per pass instead of four; penalty_reduce is the stage kernels' own fixed-order sum.
    double *mine = red + lane * 21;
#pragma unroll 1
    for (int q0 = 0; q0 < N; q0 += ppg) {
        const int npieces = min(ppg, N - q0);
        const bool exists = pl < npieces;
        const int pfl = (dp.piece_active && exists) ? dp.piece_active[p0 + q0 + pl] : DV_EVAL;
        const bool active = exists && (pfl & DV_EVAL);
        __syncthreads();                                                        // staging done (first pass) / the previous pass's partials have been summed
        if (active) {
            double o[20];
            LdsView c(cS + (q0 + pl) * 18), hb(hS + (size_t)(q0 + pl) * hstride);
            const int K = (int)hb[3], kappa = dp.kappa;
            const double step = tS[q0 + pl];
            penalty_sample_partials<true>(dp, c, hb, K, Kmax, step * jl, step, (jl == 0 || jl == kappa) ? 0.5 : 1.0, dp.inv_kappa, jl, o);
#pragma unroll
            for (int i = 0; i < 20; i++) mine[i] = o[i];
        } else if (exists) {                                                    // a piece that is switched off this round: its partials are zeros
#pragma unroll
            for (int i = 0; i < 20; i++) mine[i] = 0.0;
        }
        __syncthreads();
        penalty_reduce<false>(red, npieces, lpp, out20 + (size_t)(p0 + q0) * 20, lane, nthr, true);
    }
}

// The candidate's corridor blocks are constants of the handle: their first five 16-byte trips per thread (all of them up to K = 8 at 64 pieces) are requested at the
// kernel's ENTRY and wait in registers through the forward map (69 of the kernel's 180 VGPRs are in use there) - at the start of the penalty phase they cost a
// trip to HBM otherwise (the coefficients next to them come back from the L2 this workgroup just wrote them to).
__device__ __forceinline__ void solo_prefetch_corridor(const DevProblem &dp, int Kmax, int b, double2 (&hv)[5]) {
    const int p0 = dp.poff[b], N = dp.poff[b + 1] - p0, hstride = (Kmax + 1) * 4;
    const double2 *h2 = (const double2 *)(dp.hblk + (size_t)p0 * hstride);
    const int nh2 = (N * hstride) >> 1;
#pragma unroll
    for (int u = 0; u < 5; u++) { const int i = (int)threadIdx.x + u * 256; hv[u] = h2[i < nh2 ? i : nh2 - 1]; }
}

template <int LPP>
__global__ __launch_bounds__(256, 2) void k_eval_solo(DevProblem dp, SoloArgs a, LineSearchTap tap) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int b = blockIdx.x;
    if (dp.cand_active && !(dp.cand_active[b] & DV_EVAL)) {                    // skipped candidate: only the arrival count (k_backward_knot64)
        if (threadIdx.x == 0 && tap.arrive && atomicAdd(tap.arrive, 1u) + 1u == (unsigned)gridDim.x * tap.round) *tap.flag = tap.round;
        return;
    }
// (cycle stamps of candidate 0, frx_profile_phases with the form forced: slots 7 / 13 / 14 / 15 = entry, forward map done, penalty phase done, end - the bodies' own stamps use 0-6, 8-12 and 16-31)
#define SOLO_STAMP(slot) do { if (dp.stamps && b == 0 && threadIdx.x == 0) dp.stamps[slot] = (long long)__builtin_readcyclecounter(); } while (0)
    double2 hv[5];
    solo_prefetch_corridor(dp, a.Kmax, b, hv);
    SOLO_STAMP(7);
    forward_knot_body<false, 64>(dp, a.x, a.T, a.C, a.maxCN, a.maxXb, a.maxVb, 64, (a.dbg & 1) ? nullptr : a.pcrw, a.nsteps, b, sm);
    __syncthreads();                                                            // (C, T) and the multipliers of this candidate are out - this workgroup's own stores, read back below
    SOLO_STAMP(13);
    solo_penalty_phase<LPP>(dp, a.T, a.C, a.out20, a.lpp, a.ppg, a.Kmax, b, sm, hv);
    __syncthreads();
    SOLO_STAMP(14);
    backward_knot_body<false, 64>(dp, a.x, a.T, a.C, a.out20, a.f, a.g, a.maxCN, a.maxXb, a.maxVb, 64, a.pcrw, a.nsteps, tap, b, sm);
    SOLO_STAMP(15);
#undef SOLO_STAMP
}

} // namespace frx
