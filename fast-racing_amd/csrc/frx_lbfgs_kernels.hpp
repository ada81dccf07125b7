// Device side of the device-vector L-BFGS (SolverDV in frx_lbfgs.hpp): everything O(n) of
// lbfgs::lbfgs_optimize (lbfgs.hpp:1103-1444) — trial point x = xp + step*d, the dot products the line
// search and the convergence tests need, the (s, y) history and the two-loop recursion — runs here, one
// WORKGROUP per candidate, the search direction resident in registers (E elements per thread, n <= 64 W E).
// Dot products are reduced with DPP row operations in a fixed order (deterministic run to run).
#pragma once
#include <hip/hip_runtime.h>

#include "frx_lbfgs.hpp"
#include "frx_wave.hpp"

#ifndef FRX_KERNEL_LINKAGE
#define FRX_KERNEL_LINKAGE          // see frx_kernels.hpp
#endif

namespace frx {

struct DvBuffers {
    const int *xoff;          // [B+1]
    double *x, *g;            // packed evaluation input / output (the objective kernels read x, write g)
    double *xp, *gp, *d;      // packed
    double *S, *Y;            // [B][m][hs]  history: row = slot, natural element order, zero beyond n; thread t owns elements 2*(t + TPB*q) + {0,1} and
                              // issues unconditional 16-byte loads.  hs = TPB*E (every thread's pairs exist) or TIGHT (round 6): hs = n + 2 rounded up to 16
                              // doubles - the pairs beyond the row's end are read from its LAST pair, which lies in the zero tail (same cache line as the
                              // neighbours' loads: no traffic of their own).  The kernel is HBM-bound from ~100 candidates on and streams 4 m rows per accepted
                              // step: at the headline n = 641 a row is 656 instead of 768 doubles, 14.6 % fewer bytes.
    double *ys;               // [B][m]   y.s per slot
    int *dflags;              // [B] device copy of this round's command flags (for k_backward_knot's LineSearchTap)
    int *pflags;              // [P] the same per fine piece (for k_penalty), or null
    const int *poff;          // [B+1] fine-piece offsets (with pflags)
    double *gt;               // [B][m][4] cross products s_j . y_{j+d}, d = 1..3 (entry 3 unused), see k_lbfgs_pre
    int m, B;
    int hs;                   // row stride of S and Y in doubles (even; see above)
};

// One workgroup of W waves (TPB = 64 W threads) per candidate; E = doubles per THREAD (even), n <= TPB*E.  The host picks
// the (E, W) with the smallest padded slice TPB*E >= n: the recursion is bound by what ONE CU can stream (4 m history rows
// per advance), so padding is paid for in time (n ~ 700 at the headline size: 3 waves x 4 doubles = 768, not 1024).
// Why 4 waves with a barrier per step rather than 1 wave: the 2*bound steps of the two-loop recursion are strictly
// sequential and each consumes two fresh history rows (16 KB at n ~ 1000), so the rows have to be requested ~1.5 us
// (= 8-12 steps) ahead.  A wave's outstanding-load counter holds 63; with a quarter of the vector per wave a row
// costs each wave 2*E/2 loads, so 8 rows ahead fit, and the look-ahead buffers are 8 x 2 x E doubles of registers.
template <int E, int W, int PF, int BLK>
__global__ __launch_bounds__(64 * W) void k_lbfgs_pre(DvBuffers bf, const DvCommand *__restrict__ cmd, DvResult *__restrict__ res) {
    constexpr int Q = E / 2;                       // double2 elements per thread
    constexpr int TPB = 64 * W;
    constexpr int HS = TPB * E;                    // padded candidate slice of a history row, doubles
    constexpr int TAB = 512 + 2 * 16;              // ages -pad .. V1-1, pad < PF <= 16
    __shared__ double rhoA[TAB], alA[TAB], GA[BLK > 1 ? 4 * TAB : 1], part[2][8][6];
    __shared__ int voff[2 * TAB];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const DvCommand c = cmd[b];
    if (t == 0 && bf.dflags) bf.dflags[b] = c.flags;
    if (bf.pflags) { const int q0 = bf.poff[b], q1 = bf.poff[b + 1]; for (int q = q0 + t; q < q1; q += TPB) bf.pflags[q] = c.flags; }
    if (!(c.flags & (DV_INIT | DV_ADVANCE | DV_TRIAL | DV_RESTORE))) return;
    const int base = bf.xoff[b], n = bf.xoff[b + 1] - base;
    double *x = bf.x + base, *g = bf.g + base, *xp = bf.xp + base, *gp = bf.gp + base, *d = bf.d + base;
    // packed vectors: element i = 2*(t + TPB*q) + h; clamp the index so loads are unconditional, select afterwards
    int idx[E];
    bool ok[E];
#pragma unroll
    for (int e = 0; e < E; e++) { const int i = 2 * (t + TPB * (e >> 1)) + (e & 1); ok[e] = i < n; idx[e] = ok[e] ? i : n - 1; }
    auto load_vec = [&](const double *v, double *out) {
        double tmp[E];
#pragma unroll
        for (int e = 0; e < E; e++) tmp[e] = v[idx[e]];
#pragma unroll
        for (int e = 0; e < E; e++) out[e] = ok[e] ? tmp[e] : 0.0;
    };
    auto store_vec = [&](double *v, const double *in) {
#pragma unroll
        for (int e = 0; e < E; e++) if (ok[e]) v[idx[e]] = in[e];
    };
    int parity = 0;
    // block-wide sum in a fixed order: DPP inside each wave, the W wave totals added in wave order by every thread
    // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the history rows requested PF visits
    // ahead, and turn the pipeline back into one memory latency per step
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto block_sum_n = [&](double *v, int nv) {                         // nv <= 5 sums at the price of one barrier, in place
        if (nv == 4) {
            double w4[4] = {v[0], v[1], v[2], v[3]};
            const double q = wave_sum4_packed<false>(w4);
            if (lane < 4) part[parity][wave][lane] = q;             // lane k holds total k
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) if (k < nv) { const double w = wave_sum_dpp(v[k]); if (lane == 0) part[parity][wave][k] = w; }
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 5; k++) if (k < nv) {
            double r = part[parity][0][k];
#pragma unroll
            for (int w = 1; w < W; w++) r += part[parity][w][k];
            v[k] = r;
        }
        parity ^= 1;
    };
    auto block_sum = [&](double v) { block_sum_n(&v, 1); return v; };

    if (c.flags & DV_RESTORE) {                                        // lbfgs.hpp:1287-1288
        double tmp[E];
        load_vec(xp, tmp); store_vec(x, tmp);
        load_vec(gp, tmp); store_vec(g, tmp);
        return;
    }
    double dv[E];
    double dginit = 0.0;
    if (c.flags & DV_INIT) {                                           // d = -g; xp = x; gp = g
        double gv[E], xv[E], acc = 0.0;
        load_vec(g, gv); load_vec(x, xv);
#pragma unroll
        for (int e = 0; e < E; e++) { dv[e] = -gv[e]; acc += gv[e] * dv[e]; }
        store_vec(xp, xv); store_vec(gp, gv); store_vec(d, dv);
        dginit = block_sum(acc);
    } else if (c.flags & DV_ADVANCE) {                                 // lbfgs.hpp:1354-1411
        // Two-loop recursion, BLK pairs per reduction.  Sequentially, alpha_j needs s_j . q_j with q_j = q - sum_{newer i} alpha_i y_i;
        // for the pairs of one block that is  s_j . q  -  sum_i alpha_i (s_j . y_i)  with q taken at the block start, and the cross
        // products s_j . y_{j+d}, d = 1..BLK-1, do not depend on q: they are computed once, when pair j+d is stored (table gt).  The
        // same table serves the second loop (y_{j+d} . r picks up (alpha_j - beta_j) y_{j+d} . s_j).  So a block costs ONE block-wide
        // reduction + barrier instead of BLK, and no extra pass over the history.
        //
        // The loops are bound by how many instructions ONE wave has to issue per pair (every wave runs the whole serial chain), so
        // everything indexable is tabulated up front: pairs are addressed by AGE (0 = newest ... last = oldest), the per-pair scalars
        // (1/(y.s), cross products, alpha) live in LDS by age, and the byte offset of every row VISIT is precomputed.  Visits [0, V1)
        // walk down the ages (V1 = bound rounded up to PF), visits [V1, 2 V1) walk back up; visits past the ends re-read an end row
        // and are neutralised by a zero 1/(y.s).  Visit v lives in buffer v % PF and is requested PF visits ahead, unconditionally
        // (a conditional reload makes the compiler wait for the load at once).
        const int m = bf.m;
        const int hs = bf.hs, hs2 = hs >> 1;                            // row stride (doubles), pairs per row
        double *Sb = bf.S + (size_t)b * m * hs, *Yb = bf.Y + (size_t)b * m * hs;      // this candidate's [m][hs] history blocks
        double *ysrow = bf.ys + (size_t)b * m, *gtrow = bf.gt + (size_t)b * m * 4;
        const int jnew = c.slot, last = c.bound - 1;                   // newest pair; older pairs are jnew-1, jnew-2, ... (mod m)
        const int V1 = ((c.bound + PF - 1) / PF) * PF, pad = V1 - c.bound;               // table index = age + pad
        auto row_of_age = [&](int a) { int j = jnew - a; return j < 0 ? j + m : j; };
        for (int i = t; i < V1 + pad; i += TPB) {                      // ages -pad .. V1-1
            const int a = i - pad;
            const bool real = a >= 1 && a <= last;                     // age 0 is the pair formed below
            const int j = row_of_age(real ? a : 0);
            const double v = real ? ysrow[j] : 0.0;
            rhoA[i] = v != 0.0 ? 1.0 / v : 0.0;
            alA[i] = 0.0;
            if (BLK > 1) { GA[4 * i] = real ? gtrow[4 * j] : 0.0; GA[4 * i + 1] = real ? gtrow[4 * j + 1] : 0.0; GA[4 * i + 2] = real ? gtrow[4 * j + 2] : 0.0; }
        }
        for (int v = t; v < 2 * V1 + PF; v += TPB) {
            const int a = v < V1 ? min(v, last) : max(last - (v - V1), 0);
            voff[v] = row_of_age(a) * (int)(hs * sizeof(double));
        }
        lds_barrier();
        double sb[PF][E], yb[PF][E];
        const int toff = t * (int)sizeof(double2);
        // the LAST pair of a thread may lie beyond a tight row's end: it is read from the row's last pair (zeros) instead - one offset per thread, fixed for the kernel
        const bool last_in = t + TPB * (Q - 1) < hs2;
        const int lastoff = (last_in ? t + TPB * (Q - 1) : hs2 - 1) * (int)sizeof(double2) - toff;
        auto load_row = [&](int u, int v) {                           // buffer u <- row of visit v
            const int off = voff[v] + toff;
            const double2 *Sj = (const double2 *)((const char *)Sb + off), *Yj = (const double2 *)((const char *)Yb + off);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const double2 a = q == Q - 1 ? *(const double2 *)((const char *)Sj + lastoff) : Sj[TPB * q];
                const double2 bq = q == Q - 1 ? *(const double2 *)((const char *)Yj + lastoff) : Yj[TPB * q];
                sb[u][2 * q] = a.x; sb[u][2 * q + 1] = a.y; yb[u][2 * q] = bq.x; yb[u][2 * q + 1] = bq.y;
            }
        };
#pragma unroll
        for (int u = 1; u < PF; u++) load_row(u, u);                   // visit 0 is the pair formed below, kept in registers
        double hd[5] = {0.0, 0.0, 0.0, 0.0, 0.0};                      // y.s, y.y, and s_{age d} . y_new for d = 1..3
        {
            double xv[E], gv[E], tmp[E];
            load_vec(x, xv); load_vec(g, gv);
            load_vec(xp, tmp);
#pragma unroll
            for (int e = 0; e < E; e++) sb[0][e] = xv[e] - tmp[e];     // s = x - xp
            load_vec(gp, tmp);
#pragma unroll
            for (int e = 0; e < E; e++) { yb[0][e] = gv[e] - tmp[e]; hd[0] += yb[0][e] * sb[0][e]; hd[1] += yb[0][e] * yb[0][e]; dv[e] = -gv[e]; }
            store_vec(xp, xv); store_vec(gp, gv);                       // the accepted point becomes the base of the next search
            double2 *Sw = (double2 *)(Sb + (size_t)jnew * hs), *Yw = (double2 *)(Yb + (size_t)jnew * hs);
#pragma unroll
            for (int q = 0; q < Q; q++)
                if (q < Q - 1 || last_in) { Sw[t + TPB * q] = make_double2(sb[0][2 * q], sb[0][2 * q + 1]); Yw[t + TPB * q] = make_double2(yb[0][2 * q], yb[0][2 * q + 1]); }   // (a pair beyond a tight row's end is zero and has no place)
            if (BLK > 1) {
#pragma unroll
                for (int d = 1; d < 4 && d < PF; d++)
#pragma unroll
                    for (int e = 0; e < E; e++) hd[1 + d] += sb[d][e] * yb[0][e];
            }
        }
        block_sum_n(hd, BLK > 1 ? 5 : 2);
        const double ys = hd[0], yy = hd[1];
        if (t == 0) {
            rhoA[pad] = 1.0 / ys; ysrow[jnew] = ys;
            if (BLK > 1)
                for (int d = 1; d < 4 && d <= last; d++) { GA[4 * (pad + d) + d - 1] = hd[1 + d]; gtrow[4 * row_of_age(d) + d - 1] = hd[1 + d]; }
        }
        lds_barrier();
        for (int v0 = 0; v0 < V1; v0 += PF) {
#pragma unroll
            for (int ub = 0; ub < PF; ub += BLK) {                     // one block: visits v0+ub .. +BLK-1 = ages, buffers ub ..
                const int ia = pad + v0 + ub;                          // table index of the block's first (newest) pair
                double raw[BLK], al[BLK];
#pragma unroll
                for (int k = 0; k < BLK; k++) {
                    raw[k] = 0.0;
#pragma unroll
                    for (int e = 0; e < E; e++) raw[k] += sb[ub + k][e] * dv[e];
                }
                block_sum_n(raw, BLK);
#pragma unroll
                for (int k = 0; k < BLK; k++) {                        // alpha = (s . q) / (y . s)
                    double a = raw[k];
#pragma unroll
                    for (int i = 0; i < k; i++) a -= al[i] * GA[4 * (ia + k) + (k - i - 1)];   // pair i of the block is k-i newer than pair k
                    al[k] = a * rhoA[ia + k];
                }
                if (t == 0) {
#pragma unroll
                    for (int k = 0; k < BLK; k++) alA[ia + k] = al[k];
                }
#pragma unroll
                for (int k = 0; k < BLK; k++)
#pragma unroll
                    for (int e = 0; e < E; e++) dv[e] -= al[k] * yb[ub + k][e];
#pragma unroll
                for (int k = 0; k < BLK; k++) load_row(ub + k, v0 + ub + k + PF);
            }
        }
        lds_barrier();                                                 // alA complete before the second loop reads it
        const double h0 = ys / yy;
#pragma unroll
        for (int e = 0; e < E; e++) dv[e] *= h0;
        for (int v0 = 0; v0 < V1; v0 += PF) {
#pragma unroll
            for (int ub = 0; ub < PF; ub += BLK) {
                const int ia = pad + last - (v0 + ub);                 // table index of the block's first (oldest) pair; the next is ia-1
                double raw[BLK], cf[BLK];
#pragma unroll
                for (int k = 0; k < BLK; k++) {
                    raw[k] = 0.0;
#pragma unroll
                    for (int e = 0; e < E; e++) raw[k] += yb[ub + k][e] * dv[e];
                }
                block_sum_n(raw, BLK);
#pragma unroll
                for (int k = 0; k < BLK; k++) {                        // beta = (y . r) / (y . s);  r += (alpha - beta) s
                    double bsum = raw[k];
#pragma unroll
                    for (int i = 0; i < k; i++) bsum += cf[i] * GA[4 * (ia - i) + (k - i - 1)];   // pair k of the block is k-i newer than pair i
                    cf[k] = alA[ia - k] - bsum * rhoA[ia - k];
                }
#pragma unroll
                for (int k = 0; k < BLK; k++)
#pragma unroll
                    for (int e = 0; e < E; e++) dv[e] += cf[k] * sb[ub + k][e];
#pragma unroll
                for (int k = 0; k < BLK; k++) load_row(ub + k, min(V1 + v0 + ub + k + PF, 2 * V1 + PF - 1));
            }
        }
        double gv[E], acc = 0.0;
        load_vec(g, gv);
#pragma unroll
        for (int e = 0; e < E; e++) acc += gv[e] * dv[e];
        store_vec(d, dv);
        dginit = block_sum(acc);                                       // gp . d of the new search (lbfgs.hpp:756)
    } else {
        load_vec(d, dv);
    }
    if (c.flags & DV_TRIAL) {                                          // x = xp + step * d   (lbfgs.hpp:825-826)
        double xb[E];
        load_vec(xp, xb);
#pragma unroll
        for (int e = 0; e < E; e++) xb[e] += c.step * dv[e];
        store_vec(x, xb);
    }
    if (t == 0 && (c.flags & (DV_INIT | DV_ADVANCE))) res[b].dginit = dginit;
}

// after the objective kernels: f, g.d, x.x, g.g per candidate (lbfgs.hpp:830, 1296-1297)
FRX_KERNEL_LINKAGE __global__ __launch_bounds__(64) void k_lbfgs_post(DvBuffers bf, const double *__restrict__ f, const DvCommand *__restrict__ cmd,
                                                   DvResult *__restrict__ res) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (!(cmd[b].flags & DV_EVAL)) return;
    const int base = bf.xoff[b], n = bf.xoff[b + 1] - base;
    const double *x = bf.x + base, *g = bf.g + base, *d = bf.d + base;
    double a_dg = 0.0, a_xx = 0.0, a_gg = 0.0;
#pragma unroll 4
    for (int i = lane; i < n; i += 64) {
        const double xv = x[i], gv = g[i];
        a_dg += gv * d[i]; a_xx += xv * xv; a_gg += gv * gv;
    }
    const double dg = wave_sum_dpp(a_dg), xx = wave_sum_dpp(a_xx), gg = wave_sum_dpp(a_gg);
    if (lane == 0) { DvResult *r = res + b; r->f = f[b]; r->dg = dg; r->xx = xx; r->gg = gg; }
}

} // namespace frx
