// Device side of the device-vector L-BFGS (SolverDV in frx_lbfgs.hpp): everything O(n) of
// lbfgs::lbfgs_optimize (lbfgs.hpp:1103-1444) — trial point x = xp + step*d, the dot products the line
// search and the convergence tests need, the (s, y) history and the two-loop recursion — runs here, one
// WAVE per candidate, the search direction resident in registers (E elements per lane, n <= 64 E).
// Dot products are reduced with DPP row operations in a fixed order (deterministic run to run).
#pragma once
#include <hip/hip_runtime.h>

#include "frx_lbfgs.hpp"

namespace frx {

template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
// Sum over the 64 lanes, result broadcast to every lane (as a wave-uniform value).
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);     // row_mirror        -> every lane holds the sum of its row of 16
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3 -> row 3 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

struct DvBuffers {
    const int *xoff;          // [B+1]
    double *x, *g;            // packed evaluation input / output (the objective kernels read x, write g)
    double *xp, *gp, *d;      // packed
    double *S, *Y;            // [m][B][256*E]  history: row = slot, candidate slices zero-padded to 256*E doubles, so every
                              // thread can issue unconditional 16-byte loads (thread t owns elements 2*(t + 256*q) + {0,1})
    double *ys;               // [B][m]   y.s per slot
    int m, B;
};

// One 256-thread workgroup (4 waves) per candidate; E = doubles per THREAD (even), n <= 256*E.
// Why 4 waves with a barrier per step rather than 1 wave: the 2*bound steps of the two-loop recursion are strictly
// sequential and each consumes two fresh history rows (16 KB at n ~ 1000), so the rows have to be requested ~1.5 us
// (= 8-12 steps) ahead.  A wave's outstanding-load counter holds 63; with a quarter of the vector per wave a row
// costs each wave 2*E/2 loads, so 8 rows ahead fit, and the look-ahead buffers are 8 x 2 x E doubles of registers.
template <int E>
__global__ __launch_bounds__(256) void k_lbfgs_pre(DvBuffers bf, const DvCommand *__restrict__ cmd, DvResult *__restrict__ res) {
    constexpr int Q = E / 2;                       // double2 elements per thread
    constexpr int HS = 256 * E;                    // padded candidate slice of a history row, doubles
    __shared__ double rhoS[512], alS[512], part[2][4];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const DvCommand c = cmd[b];
    if (!(c.flags & (DV_INIT | DV_ADVANCE | DV_TRIAL | DV_RESTORE))) return;
    const int base = bf.xoff[b], n = bf.xoff[b + 1] - base;
    double *x = bf.x + base, *g = bf.g + base, *xp = bf.xp + base, *gp = bf.gp + base, *d = bf.d + base;
    // packed vectors: element i = 2*(t + 256*q) + h; clamp the index so loads are unconditional, select afterwards
    int idx[E];
    bool ok[E];
#pragma unroll
    for (int e = 0; e < E; e++) { const int i = 2 * (t + 256 * (e >> 1)) + (e & 1); ok[e] = i < n; idx[e] = ok[e] ? i : n - 1; }
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
    // block-wide sum in a fixed order: DPP inside each wave, the four wave totals added in wave order by every thread
    auto block_sum = [&](double v) {
        const double w = wave_sum_dpp(v);
        if (lane == 0) part[parity][wave] = w;
        // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the history rows requested PF steps
        // ahead, and turn the pipeline back into one memory latency per step
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const double r = ((part[parity][0] + part[parity][1]) + part[parity][2]) + part[parity][3];
        parity ^= 1;
        return r;
    };

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
        const int m = bf.m;
        const size_t rowstride = (size_t)bf.B * HS;                    // doubles between consecutive slots
        double *Sb = bf.S + (size_t)b * HS, *Yb = bf.Y + (size_t)b * HS;
        double *ysrow = bf.ys + (size_t)b * m;
        for (int j = t; j < m; j += 256) { const double v = ysrow[j]; rhoS[j] = v != 0.0 ? 1.0 / v : 0.0; }
        double a_ys = 0.0, a_yy = 0.0;
        {
            double xv[E], gv[E], tmp[E], sv[E], yv[E];
            load_vec(x, xv); load_vec(g, gv);
            load_vec(xp, tmp);
#pragma unroll
            for (int e = 0; e < E; e++) sv[e] = xv[e] - tmp[e];         // s = x - xp
            load_vec(gp, tmp);
#pragma unroll
            for (int e = 0; e < E; e++) { yv[e] = gv[e] - tmp[e]; a_ys += yv[e] * sv[e]; a_yy += yv[e] * yv[e]; dv[e] = -gv[e]; }
            store_vec(xp, xv); store_vec(gp, gv);                       // the accepted point becomes the base of the next search
            double2 *Sw = (double2 *)(Sb + (size_t)c.slot * rowstride), *Yw = (double2 *)(Yb + (size_t)c.slot * rowstride);
#pragma unroll
            for (int q = 0; q < Q; q++) { Sw[t + 256 * q] = make_double2(sv[2 * q], sv[2 * q + 1]); Yw[t + 256 * q] = make_double2(yv[2 * q], yv[2 * q + 1]); }
        }
        const double ys = block_sum(a_ys), yy = block_sum(a_yy);
        if (t == 0) { rhoS[c.slot] = 1.0 / ys; ysrow[c.slot] = ys; }
        __syncthreads();
        constexpr int PF = E <= 4 ? 8 : 4;
        double sb[PF][E], yb[PF][E];
        const int jnew = c.slot;                                       // newest pair; older pairs are jnew-1, jnew-2, ... (mod m)
        auto row_of_down = [&](int it) { int j = jnew - it; return j < 0 ? j + m : j; };            // first loop: newest -> oldest
        auto load_row = [&](int u, int j) {
            const double2 *Sj = (const double2 *)(Sb + (size_t)j * rowstride), *Yj = (const double2 *)(Yb + (size_t)j * rowstride);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const double2 a = Sj[t + 256 * q], bq = Yj[t + 256 * q];
                sb[u][2 * q] = a.x; sb[u][2 * q + 1] = a.y; yb[u][2 * q] = bq.x; yb[u][2 * q + 1] = bq.y;
            }
        };
        // the pair written above is read back by other threads' loads only through the same thread's own elements
        // (thread t reads exactly what thread t wrote), so no device-scope fence is needed before the first loop
        // Branch-free groups of PF steps (loads unconditional, row index clamped into the valid range): a conditional
        // reload would make the compiler merge old/new buffer registers and wait for the load it has just issued.
        const int last = c.bound - 1, nfull = (c.bound / PF) * PF;
        auto step_down = [&](int it, int u, bool refill) {
            const int j = row_of_down(it);
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < E; e++) acc += sb[u][e] * dv[e];
            const double a = block_sum(acc) * rhoS[j];                 // alpha_j = (s_j . q) / (y_j . s_j)
            if (t == 0) alS[j] = a;
#pragma unroll
            for (int e = 0; e < E; e++) dv[e] -= a * yb[u][e];
            if (refill) load_row(u, row_of_down(min(it + PF, last)));
        };
#pragma unroll
        for (int u = 0; u < PF; u++) load_row(u, row_of_down(min(u, last)));
        for (int it0 = 0; it0 < nfull; it0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; u++) step_down(it0 + u, u, true);
        }
#pragma unroll
        for (int u = 0; u < PF; u++) if (nfull + u < c.bound) step_down(nfull + u, u, false);
        __syncthreads();
        const double h0 = ys / yy;
#pragma unroll
        for (int e = 0; e < E; e++) dv[e] *= h0;
        const int jold = row_of_down(c.bound - 1);                     // oldest pair in use
        auto row_of_up = [&](int it) { int j = jold + it; return j >= m ? j - m : j; };              // second loop: oldest -> newest
        auto step_up = [&](int it, int u, bool refill) {
            const int j = row_of_up(it);
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < E; e++) acc += yb[u][e] * dv[e];
            const double coef = alS[j] - block_sum(acc) * rhoS[j];     // alpha_j - beta_j
#pragma unroll
            for (int e = 0; e < E; e++) dv[e] += coef * sb[u][e];
            if (refill) load_row(u, row_of_up(min(it + PF, last)));
        };
#pragma unroll
        for (int u = 0; u < PF; u++) load_row(u, row_of_up(min(u, last)));
        for (int it0 = 0; it0 < nfull; it0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; u++) step_up(it0 + u, u, true);
        }
#pragma unroll
        for (int u = 0; u < PF; u++) if (nfull + u < c.bound) step_up(nfull + u, u, false);
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
__global__ __launch_bounds__(64) void k_lbfgs_post(DvBuffers bf, const double *__restrict__ f, const DvCommand *__restrict__ cmd,
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
