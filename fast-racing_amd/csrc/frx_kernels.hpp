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
//   hrec   [sumK][6]    half-space records (unit normal, point), CSR by polytope
//   out20  [P][20]      per-piece partials {cost, gdT, gdC[6][3]} written by k_penalty
//   band   [sum 6N_b*13] LU factors per candidate, row-window layout A(i,j) → [i*13 + (j-i+6)]
//   x, g   packed per candidate (tau[dimT], xi[...]);  f [B]
#pragma once
#include <hip/hip_runtime.h>

#include "frx_device.hpp"

namespace frx {


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
__global__ __launch_bounds__(64) void k_forward(DevProblem dp, const double *__restrict__ x, double *__restrict__ Tout,
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
// k_penalty: block = 256 (4 waves); each wave owns ppw = 64/lpp consecutive pieces, lpp =
// min(kappa+1, 64) lanes per piece, lane = one quadrature sample (strided when kappa+1 > 64).
// Dynamic LDS: cS[ppb*18] | tS[ppb] | hsS[ppb*Kmax*6] | red[4][64*21]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_penalty(DevProblem dp, const double *__restrict__ T, const double *__restrict__ C,
                                                 double *__restrict__ out20, int lpp, int ppw, int Kmax) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ppb = 4 * ppw;
    const int gp0 = blockIdx.x * ppb;
    const int npieces = min(ppb, dp.P - gp0);
    double *cS = sm;
    double *tS = cS + ppb * 18;
    double *hsS = tS + ppb;
    double *red = hsS + (size_t)ppb * Kmax * 6 + (size_t)wave * 64 * 21;

    // stage coefficients, durations and corridor polytopes (contiguous, coalesced reads)
    for (int i = tid; i < npieces * 18; i += 256) cS[i] = C[(size_t)gp0 * 18 + i];
    for (int i = tid; i < npieces; i += 256) tS[i] = T[gp0 + i];
    const int recs = Kmax * 6;
    for (int i = tid; i < npieces * recs; i += 256) {
        const int p = i / recs, r = i - p * recs;
        const int gp = gp0 + p;
        if (r < dp.piece_K[gp] * 6) hsS[i] = dp.hrec[(size_t)dp.piece_hbeg[gp] * 6 + r];
    }
    __syncthreads();

    const int pl = lane / lpp, jl = lane - pl * lpp;
    const int p = wave * ppw + pl;
    const bool active = pl < ppw && p < npieces;
    double acc[20];
#pragma unroll
    for (int v = 0; v < 20; v++) acc[v] = 0.0;
    if (active) {
        const int gp = gp0 + p;
        const double *c = cS + p * 18;
        const double *hs = hsS + (size_t)p * recs;
        const int K = dp.piece_K[gp];
        const int kappa = dp.kappa;
        const double step = tS[p] / kappa;                       // CPU.hpp:245
        const double invK = 1.0 / kappa;
        for (int j = jl; j <= kappa; j += lpp) {
            const double s1 = step * j;                           // sample abscissa as cc.cu:152
            const double omg = (j == 0 || j == kappa) ? 0.5 : 1.0;   // CPU.hpp:306
            const double alpha = invK * j;                        // CPU.hpp:259
            double adj[12], Ps, gTa;
            penalty_sample(c, s1, omg * step, dp.pc, hs, K, adj, Ps, gTa);
            const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
            acc[0] += omg * step * Ps;
            acc[1] += alpha * gTa + omg * Ps / kappa;
            const double b0[6] = {1.0, s1, s2, s3, s4, s5};
            const double b1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
            const double b2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
            const double b3[6] = {0.0, 0.0, 0.0, 6.0, 24.0 * s1, 60.0 * s2};
#pragma unroll
            for (int k = 0; k < 6; k++)
#pragma unroll
                for (int d = 0; d < 3; d++)
                    acc[2 + 3 * k + d] += b0[k] * adj[d] + b1[k] * adj[3 + d] + b2[k] * adj[6 + d] + b3[k] * adj[9 + d];
        }
    }
#pragma unroll
    for (int v = 0; v < 20; v++) red[lane * 21 + v] = acc[v];
    __syncthreads();
    // fixed-order reduction over the samples of each piece: lane = (piece-in-wave, value)
    for (int idx = lane; idx < ppw * 20; idx += 64) {
        const int p2 = idx / 20, v = idx - p2 * 20;
        const int pp = wave * ppw + p2;
        if (pp < npieces) {
            double s = 0.0;
            for (int l = 0; l < lpp; l++) s += red[(p2 * lpp + l) * 21 + v];
            out20[(size_t)(gp0 + pp) * 20 + v] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_backward: grid = B, block = 64.  Dynamic LDS: band[6N*13] | gd[6N*3] | cL[6N*3] | Tf[N] | gT[N] | gC[cN]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_backward(DevProblem dp, const double *__restrict__ x, const double *__restrict__ Tin,
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

#undef BAND
} // namespace frx
