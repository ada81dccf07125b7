// MINCO minimum-jerk spline map (q, T) -> c and its adjoint, in a form with O(log N) depth.
//
// The reference solves the 6N x 6N banded system A(T) c = b(q) by no-pivot LU (CPU.hpp:425-505,
// traj.hpp:655-719) and back-propagates with A^-T (traj.hpp:724-751) + addPropCtoT/P (CPU.hpp:104-161):
// five sweeps of 6N strictly sequential rows per evaluation — the latency floor of a GPU
// evaluation (measured: 346 of 352 us at N = 64).
//
// Rows of A say: head/tail PVA, waypoint interpolation, and continuity of p, v, a, JERK and SNAP at
// every interior knot (CPU.hpp:446-476).  Eliminating what is local gives the classical form:
// piece i is the quintic Hermite interpolant of (p,v,a) at its two knots, and the only coupled
// unknowns are w_k = (v_k, a_k) at the interior knots k = 1..N-1, fixed by jerk/snap continuity:
//       L_k w_{k-1} + D_k w_k + U_k w_{k+1} = r_k          (2x2 blocks, 3 right-hand sides = axes)
// Writing the two rows as (-(snap jump), +(jerk jump)) makes the matrix the (half) Hessian of the
// jerk energy with respect to the free knot derivatives: symmetric positive definite, U_k = L_{k+1}^T.
// It is solved by parallel cyclic reduction (one lane per knot, ceil(log2(N-1)) steps), which is
// stable for SPD block-tridiagonal systems.  Same spline, same exact-arithmetic result as the
// reference's LU; FP64 rounding differs at the 1e-13 level (tests compare with the oracle).
//
// Adjoint (replaces A^-T, addPropCtoT, addPropCtoP): with cbar = d f / d c,
//   1. piece-local:   dbar = (dc/dd)^T cbar,  hbar_local = cbar . dc/dh          (d = knot data of the piece)
//   2. knot-wise:     wbar_k = contributions of the two adjacent pieces
//   3. mu = K^-1 wbar (K symmetric: the SAME reduction with another right-hand side)
//   4. pbar_k  += sum_j mu_j . d r_j / d p_k ,   hbar_i += - mu . d(K w - r)/d h_i
#pragma once
#include "frx_math.hpp"

namespace frx {

// 1/x.  Device: v_rcp_f64 refined by two Newton steps (5 instructions on the critical path of every reduction step instead of the
// ~10 of an IEEE division with its scale / fixup sequence); relative error < 2^-51.  Host: the plain division.
FRX_HD double rcp_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
#else
    return 1.0 / x;
#endif
}

// One block row of the knot system: D (2x2), L (2x2, couples to knot k-s), U (2x2, knot k+s), r (2 x 3 axes).
struct KnotRow {
    double D[4], L[4], U[4], r[6];   // row-major 2x2; r[row*3 + axis]
};

FRX_HD void knot_row_identity(KnotRow &R) {
    R.D[0] = 1.0; R.D[1] = 0.0; R.D[2] = 0.0; R.D[3] = 1.0;
    for (int i = 0; i < 4; i++) { R.L[i] = 0.0; R.U[i] = 0.0; }
    for (int i = 0; i < 6; i++) R.r[i] = 0.0;
}

// Matrix part of knot k between a left piece of duration hL and a right piece of duration hR.
FRX_HD void knot_row_matrix(double hL, double hR, KnotRow &R) {
    const double iL = rcp_fast(hL), iL2 = iL * iL, iL3 = iL2 * iL;
    const double iR = rcp_fast(hR), iR2 = iR * iR, iR3 = iR2 * iR;
    R.D[0] = 192.0 * iL3 + 192.0 * iR3; R.D[1] = -36.0 * iL2 + 36.0 * iR2;
    R.D[2] = R.D[1];                    R.D[3] = 9.0 * iL + 9.0 * iR;
    R.L[0] = 168.0 * iL3; R.L[1] = 24.0 * iL2; R.L[2] = -24.0 * iL2; R.L[3] = -3.0 * iL;
    R.U[0] = 168.0 * iR3; R.U[1] = -24.0 * iR2; R.U[2] = 24.0 * iR2; R.U[3] = -3.0 * iR;
}
// Right-hand side of knot k for one axis from the position differences dL = p_k - p_{k-1}, dR = p_{k+1} - p_k.
FRX_HD void knot_row_rhs(double hL, double hR, double dL, double dR, double &rv, double &ra) {
    const double iL = rcp_fast(hL), iL3 = iL * iL * iL, iR = rcp_fast(hR), iR3 = iR * iR * iR;
    rv = 360.0 * dL * iL3 * iL + 360.0 * dR * iR3 * iR;
    ra = -60.0 * dL * iL3 + 60.0 * dR * iR3;
}

// 2x2 helpers (row-major)
FRX_HD void m2_inv(const double *A, double *I) {
    const double det = A[0] * A[3] - A[1] * A[2], id = 1.0 / det;
    I[0] = A[3] * id; I[1] = -A[1] * id; I[2] = -A[2] * id; I[3] = A[0] * id;
}
// the same with rcp_fast (the wave-specialised reduction of the device, where the inversion sits on the critical path of a step)
FRX_HD void m2_inv_fast(const double *A, double *I) {
    const double det = A[0] * A[3] - A[1] * A[2], id = rcp_fast(det);
    I[0] = A[3] * id; I[1] = -A[1] * id; I[2] = -A[2] * id; I[3] = A[0] * id;
}
FRX_HD void m2_mul(const double *A, const double *B, double *C) {
    C[0] = A[0] * B[0] + A[1] * B[2]; C[1] = A[0] * B[1] + A[1] * B[3];
    C[2] = A[2] * B[0] + A[3] * B[2]; C[3] = A[2] * B[1] + A[3] * B[3];
}

// One parallel-cyclic-reduction step for row `me` with its neighbours at distance s (`lo` = k-s, `hi` = k+s;
// pass identity rows beyond the ends).  After ceil(log2(n)) steps with s = 1,2,4,.. every row is decoupled.
// al = L D_lo^-1 and be = U D_hi^-1 are the only things a further right-hand side needs from this step
// (pcr_rhs_step): the adjoint solve of the same evaluation re-uses them instead of reducing the matrix again.
FRX_HD void pcr_step(const KnotRow &me, const KnotRow &lo, const KnotRow &hi, KnotRow &out, double *al, double *be) {
    double iLo[4], iHi[4], t[4];
    m2_inv(lo.D, iLo);
    m2_inv(hi.D, iHi);
    m2_mul(me.L, iLo, al);                 // alpha = L D_lo^-1   (applied with a minus sign below)
    m2_mul(me.U, iHi, be);                 // beta  = U D_hi^-1
    m2_mul(al, lo.U, t);
    for (int i = 0; i < 4; i++) out.D[i] = me.D[i] - t[i];
    m2_mul(be, hi.L, t);
    for (int i = 0; i < 4; i++) out.D[i] -= t[i];
    m2_mul(al, lo.L, t);
    for (int i = 0; i < 4; i++) out.L[i] = -t[i];
    m2_mul(be, hi.U, t);
    for (int i = 0; i < 4; i++) out.U[i] = -t[i];
    for (int a = 0; a < 3; a++) {
        out.r[a] = me.r[a] - (al[0] * lo.r[a] + al[1] * lo.r[3 + a]) - (be[0] * hi.r[a] + be[1] * hi.r[3 + a]);
        out.r[3 + a] = me.r[3 + a] - (al[2] * lo.r[a] + al[3] * lo.r[3 + a]) - (be[2] * hi.r[a] + be[3] * hi.r[3 + a]);
    }
}
// The same with the neighbours' diagonal blocks already inverted (lo.D, hi.D hold D^-1): the device row buffer stores the inverse.
FRX_HD void pcr_step_inv(const KnotRow &me, const KnotRow &lo, const KnotRow &hi, KnotRow &out, double *al, double *be) {
    double t[4];
    m2_mul(me.L, lo.D, al);
    m2_mul(me.U, hi.D, be);
    m2_mul(al, lo.U, t);
    for (int i = 0; i < 4; i++) out.D[i] = me.D[i] - t[i];
    m2_mul(be, hi.L, t);
    for (int i = 0; i < 4; i++) out.D[i] -= t[i];
    m2_mul(al, lo.L, t);
    for (int i = 0; i < 4; i++) out.L[i] = -t[i];
    m2_mul(be, hi.U, t);
    for (int i = 0; i < 4; i++) out.U[i] = -t[i];
    for (int a = 0; a < 3; a++) {
        out.r[a] = me.r[a] - (al[0] * lo.r[a] + al[1] * lo.r[3 + a]) - (be[0] * hi.r[a] + be[1] * hi.r[3 + a]);
        out.r[3 + a] = me.r[3 + a] - (al[2] * lo.r[a] + al[3] * lo.r[3 + a]) - (be[2] * hi.r[a] + be[3] * hi.r[3 + a]);
    }
}
FRX_HD void pcr_step(const KnotRow &me, const KnotRow &lo, const KnotRow &hi, KnotRow &out) {
    double al[4], be[4];
    pcr_step(me, lo, hi, out, al, be);
}
// right-hand-side part of a step alone: r (2 x 3 axes) of this row and of its two neighbours
FRX_HD void pcr_rhs_step(double *r, const double *al, const double *be, const double *rlo, const double *rhi) {
    for (int a = 0; a < 3; a++) {
        const double r0 = r[a] - (al[0] * rlo[a] + al[1] * rlo[3 + a]) - (be[0] * rhi[a] + be[1] * rhi[3 + a]);
        const double r1 = r[3 + a] - (al[2] * rlo[a] + al[3] * rlo[3 + a]) - (be[2] * rhi[a] + be[3] * rhi[3 + a]);
        r[a] = r0; r[3 + a] = r1;
    }
}
// decoupled row: w = D^-1 r  -> (v[3], a[3])
FRX_HD void pcr_finish(const KnotRow &R, double *v, double *a) {
    double I[4];
    m2_inv(R.D, I);
    for (int x = 0; x < 3; x++) {
        v[x] = I[0] * R.r[x] + I[1] * R.r[3 + x];
        a[x] = I[2] * R.r[x] + I[3] * R.r[3 + x];
    }
}

// Quintic Hermite coefficients of one axis: c[k] = coefficient of t^k on [0, h].
FRX_HD void hermite_coeffs(double h, double p0, double v0, double a0, double p1, double v1, double a1, double *c) {
    const double ih = rcp_fast(h), ih2 = ih * ih, ih3 = ih2 * ih, ih4 = ih2 * ih2, ih5 = ih4 * ih;
    const double dl = p1 - p0;
    c[0] = p0; c[1] = v0; c[2] = 0.5 * a0;
    c[3] = 10.0 * dl * ih3 - (4.0 * v1 + 6.0 * v0) * ih2 - 0.5 * (3.0 * a0 - a1) * ih;
    c[4] = -15.0 * dl * ih4 + (7.0 * v1 + 8.0 * v0) * ih3 + 0.5 * (3.0 * a0 - 2.0 * a1) * ih2;
    c[5] = 6.0 * dl * ih5 - 3.0 * (v1 + v0) * ih4 - 0.5 * (a0 - a1) * ih3;
}
// Adjoint of hermite_coeffs for one axis: cb[6] -> (p0b, v0b, a0b, p1b, v1b, a1b) and the duration adjoint.
FRX_HD void hermite_adjoint(double h, double p0, double v0, double a0, double p1, double v1, double a1, const double *cb,
                            double *db, double &hb) {
    const double ih = rcp_fast(h), ih2 = ih * ih, ih3 = ih2 * ih, ih4 = ih2 * ih2, ih5 = ih4 * ih, ih6 = ih3 * ih3;
    const double dl = p1 - p0;
    const double pd = 10.0 * ih3 * cb[3] - 15.0 * ih4 * cb[4] + 6.0 * ih5 * cb[5];
    db[0] = cb[0] - pd;
    db[3] = pd;
    db[1] = cb[1] - 6.0 * ih2 * cb[3] + 8.0 * ih3 * cb[4] - 3.0 * ih4 * cb[5];
    db[4] = -4.0 * ih2 * cb[3] + 7.0 * ih3 * cb[4] - 3.0 * ih4 * cb[5];
    db[2] = 0.5 * cb[2] - 1.5 * ih * cb[3] + 1.5 * ih2 * cb[4] - 0.5 * ih3 * cb[5];
    db[5] = 0.5 * ih * cb[3] - ih2 * cb[4] + 0.5 * ih3 * cb[5];
    const double dc3 = -30.0 * dl * ih4 + 2.0 * (4.0 * v1 + 6.0 * v0) * ih3 + 0.5 * (3.0 * a0 - a1) * ih2;
    const double dc4 = 60.0 * dl * ih5 - 3.0 * (7.0 * v1 + 8.0 * v0) * ih4 - (3.0 * a0 - 2.0 * a1) * ih3;
    const double dc5 = -30.0 * dl * ih6 + 12.0 * (v1 + v0) * ih5 + 1.5 * (a0 - a1) * ih4;
    hb = cb[3] * dc3 + cb[4] * dc4 + cb[5] * dc5;
}

// Adjoint of the knot system with respect to piece i (duration h, knots i and i+1), one axis.
//   mu0 = (mu_v, mu_a) of knot i (zero if knot i is the fixed head), mu1 of knot i+1 (zero if fixed tail)
//   returns d f / d(p_{i+1} - p_i) through the right-hand sides, and adds the duration term.
FRX_HD double knot_adjoint_piece(double h, double dl, double v0, double a0, double v1, double a1, double mu0v, double mu0a,
                                 double mu1v, double mu1a, double &hb) {
    const double ih = rcp_fast(h), ih2 = ih * ih, ih3 = ih2 * ih, ih4 = ih2 * ih2, ih5 = ih4 * ih;
    // rows of knot i+1 contributed by this piece (as its LEFT piece): (-snap(h), +jerk(h))
    const double dLv = -504.0 * v0 * ih4 - 48.0 * a0 * ih3 - 576.0 * v1 * ih4 + 72.0 * a1 * ih3 + 1440.0 * dl * ih5;
    const double dLa = 48.0 * v0 * ih3 + 3.0 * a0 * ih2 + 72.0 * v1 * ih3 - 9.0 * a1 * ih2 - 180.0 * dl * ih4;
    // rows of knot i contributed by this piece (as its RIGHT piece): (+snap(0), -jerk(0))
    const double dRv = 1440.0 * dl * ih5 - 576.0 * v0 * ih4 - 504.0 * v1 * ih4 - 72.0 * a0 * ih3 + 48.0 * a1 * ih3;
    const double dRa = 180.0 * dl * ih4 - 72.0 * v0 * ih3 - 48.0 * v1 * ih3 - 9.0 * a0 * ih2 + 3.0 * a1 * ih2;
    hb -= mu1v * dLv + mu1a * dLa + mu0v * dRv + mu0a * dRa;
    return (mu1v + mu0v) * 360.0 * ih4 + (mu0a - mu1a) * 60.0 * ih3;
}

} // namespace frx
