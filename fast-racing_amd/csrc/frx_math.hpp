// Per-sample arithmetic of the SE(3) penalty integrand, shared by the HIP kernels and (for
// CPU-side checking of the algebra only) by tests/hostcheck.  FP64 throughout.
//
// What the reference does per sample (CPU.hpp:250-398): quintic p/v/a/j/s, flat-output
// attitude R = [xB yB zB] from h = a + g e3 with its Jacobians dxB/dyB/dzB (3x3 each), body
// rate from R^T j / |h| with two more 3x3 Jacobians, then per half-space three (6x3)(3x3)
// products (CPU.hpp:318-320) and per limit a 6x3 outer product.
//
// What this file does instead: reverse-mode differentiation of the same scalar penalty.  Every
// contribution to the 6x3 coefficient gradient is  beta_m (x) a_m  with a_m in R^3 the adjoint of
// (pos, vel, acc, jer) — see SURVEY.md §7.2 — so a sample is reduced to four 3-vectors:
//     a0 = d pen / d pos      a1 = d pen / d vel      a2 = d pen / d acc      a3 = d pen / d jer
// (all pre-multiplied by the quadrature weight omega*step), plus the un-weighted penalty sum P.
//     gdC_i  += beta0 a0^T + beta1 a1^T + beta2 a2^T + beta3 a3^T              (CPU.hpp:341,354,368,381,394)
//     gdT_i  += alpha (a0.v + a1.a + a2.j + a3.s) + omega P / kappa             (CPU.hpp:342,355,369,382,395)
//     cost   += omega step P                                                     (CPU.hpp:343,357,371,384,397)
// The frame Jacobians are never formed: with G(x) = (I - x^ x^^T)/|x| (normalizeFDF, CPU.hpp:163-185)
//     dzB = G(h),   dyB = G(czB) * cdzB,   dxB[:,q] = dyB[:,q] x zB + yB x dzB[:,q]    (CPU.hpp:268-278)
// and for any weights U0,U1,U2 in R^3
//     dxB^T U0 + dyB^T U1 + dzB^T U2 = G(h) ( (0, -P'z, P'y) + U2 + U0 x yB ),  P' = G(czB)(U1 + zB x U0).
// This is a re-association of the reference's sums (relative differences ~1e-15 per sample),
// which the 1e-6 contract on optimised coefficients allows.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define FRX_HD __host__ __device__ __forceinline__
#else
#define FRX_HD inline
#endif

namespace frx {

struct PenaltyConst {         // scalar arguments of addTimeIntPenalty (CPU.hpp:188-201), squared where the reference squares them
    double ell[3];            // ellipsoid semi-axes (horiz, horiz, vert), CPU.hpp:1155-1157
    double safeMargin;
    double vMaxSqr, thrMinSqr, thrMaxSqr, bdrMaxSqr;   // CPU.hpp:204-207
    double gAcc;
    double chi[4];            // PenaltyPVTB
};

FRX_HD double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
FRX_HD void cross3(const double *a, const double *b, double *r) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}

// One quadrature sample of piece coefficients c[k*3+d] (k = power, d = axis) at local time s1.
//   ws    = omega * step  (trapezoid weight x step, CPU.hpp:245,306)
//   hs    = K half-space records (n_x,n_y,n_z,p_x,p_y,p_z), unit normals
//   adj   = out: a0,a1,a2,a3 (12 doubles), already weighted by ws
//   Psum  = out: sum of chi*viol^3 (not weighted)
//   gTalpha = out: a0.v + a1.a + a2.j + a3.s  (to be multiplied by alpha = j/kappa)
FRX_HD void penalty_sample(const double *c, double s1, double ws, const PenaltyConst &pc,
                           const double *hs, int K, double *adj, double &Psum, double &gTalpha) {
    const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2;
    // beta rows as in CPU.hpp:254-258; beta0[5] = s5 is only needed for pos
    const double s5 = s4 * s1;
    double pos[3], vel[3], acc[3], jer[3], sna[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const double c0 = c[d], c1 = c[3 + d], c2 = c[6 + d], c3 = c[9 + d], c4 = c[12 + d], c5 = c[15 + d];
        pos[d] = c0 + c1 * s1 + c2 * s2 + c3 * s3 + c4 * s4 + c5 * s5;
        vel[d] = c1 + c2 * (2.0 * s1) + c3 * (3.0 * s2) + c4 * (4.0 * s3) + c5 * (5.0 * s4);
        acc[d] = c2 * 2.0 + c3 * (6.0 * s1) + c4 * (12.0 * s2) + c5 * (20.0 * s3);
        jer[d] = c3 * 6.0 + c4 * (24.0 * s1) + c5 * (60.0 * s2);
        sna[d] = c4 * 24.0 + c5 * (120.0 * s1);
    }

    // attitude from differential flatness (CPU.hpp:266-279)
    const double h[3] = {acc[0], acc[1], acc[2] + pc.gAcc};
    const double f2 = dot3(h, h);
    const double fThr = sqrt(f2);
    const double invF = 1.0 / fThr;
    const double zB[3] = {h[0] * invF, h[1] * invF, h[2] * invF};
    const double m2 = zB[2] * zB[2] + zB[1] * zB[1];
    const double mN = sqrt(m2);
    const double invM = 1.0 / mN;
    const double yB[3] = {0.0, zB[2] * invM, -zB[1] * invM};
    double xB[3];
    cross3(yB, zB, xB);

    // body rate (CPU.hpp:285-292)
    const double r0 = dot3(xB, jer), r1 = dot3(yB, jer);
    const double sqrMagThr = fThr * fThr;
    const double b0 = r0 * invF, b1 = r1 * invF;
    const double sqrMagBdr = b1 * b1 + b0 * b0;

    const double violaVel = dot3(vel, vel) - pc.vMaxSqr;       // CPU.hpp:301-304
    const double violaThrl = pc.thrMinSqr - sqrMagThr;
    const double violaThrh = sqrMagThr - pc.thrMaxSqr;
    const double violaBdr = sqrMagBdr - pc.bdrMaxSqr;

    double a0[3] = {0, 0, 0};
    double U0[3] = {0, 0, 0}, U1[3] = {0, 0, 0}, U2[3] = {0, 0, 0};
    double P = 0.0;
    bool needReverse = false;

    // corridor half-spaces (CPU.hpp:310-345); the sign test avoids the sqrt unless violated
    const double e0 = pc.ell[0], e1 = pc.ell[1], e2 = pc.ell[2];
    double Pcorr = 0.0;
    for (int k = 0; k < K; k++) {
        const double *rec = hs + 6 * k;
        const double n[3] = {rec[0], rec[1], rec[2]};
        const double dp[3] = {pos[0] - rec[3], pos[1] - rec[4], pos[2] - rec[5]};
        const double w0 = dot3(xB, n) * e0, w1 = dot3(yB, n) * e1, w2 = dot3(zB, n) * e2;   // (R^T n) .* ellipsoid
        const double eN2 = w0 * w0 + w1 * w1 + w2 * w2;
        const double nd = dot3(n, dp);
        const double d0 = nd + pc.safeMargin;
        if (d0 >= 0.0 || eN2 > d0 * d0) {
            const double eNorm = sqrt(eN2);
            const double sd = (nd + eNorm) + pc.safeMargin;       // CPU.hpp:325,328
            if (sd > 0.0) {
                const double sd2 = sd * sd;
                const double cw = ws * pc.chi[0] * 3.0 * sd2;
                const double ie = 1.0 / eNorm;
                const double g0 = w0 * ie * e0, g1 = w1 * ie * e1, g2 = w2 * ie * e2;   // eNormGd, CPU.hpp:324,326
                const double cg0 = cw * g0, cg1 = cw * g1, cg2 = cw * g2;
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    a0[d] += cw * n[d];
                    U0[d] += cg0 * n[d];
                    U1[d] += cg1 * n[d];
                    U2[d] += cg2 * n[d];
                }
                Pcorr += sd * sd2;
                needReverse = true;
            }
        }
    }
    P += pc.chi[0] * Pcorr;

    double a1[3] = {0, 0, 0}, a2[3] = {0, 0, 0}, a3[3] = {0, 0, 0};
    if (violaVel > 0.0) {                                       // CPU.hpp:347-359
        const double v2 = violaVel * violaVel;
        const double wV = ws * pc.chi[1] * 3.0 * v2 * 2.0;
        a1[0] = wV * vel[0]; a1[1] = wV * vel[1]; a1[2] = wV * vel[2];
        P += pc.chi[1] * (v2 * violaVel);
    }
    double wH = 0.0;                                            // weight on dSqrMagThr = 2h
    if (violaThrl > 0.0) {                                      // CPU.hpp:361-372
        const double v2 = violaThrl * violaThrl;
        wH -= ws * pc.chi[2] * 3.0 * v2;
        P += pc.chi[2] * (v2 * violaThrl);
    }
    if (violaThrh > 0.0) {                                      // CPU.hpp:374-385
        const double v2 = violaThrh * violaThrh;
        wH += ws * pc.chi[2] * 3.0 * v2;
        P += pc.chi[2] * (v2 * violaThrh);
    }
    if (violaBdr > 0.0) {                                       // CPU.hpp:387-398
        const double v2 = violaBdr * violaBdr;
        const double wB = ws * pc.chi[3] * 3.0 * v2;
        const double k2 = wB * 2.0 * invF * invF;
        // d(omega^2)/d jer = (2/f^2)(r0 xB + r1 yB)   (= dJerSqrMagBdr, CPU.hpp:297-299)
#pragma unroll
        for (int d = 0; d < 3; d++) a3[d] = k2 * (r0 * xB[d] + r1 * yB[d]);
        // d(omega^2)/d h = (2/f^2)(r0 dxB^T jer + r1 dyB^T jer) - 2 omega^2 h / f^2   (= dSqrMagBdr, CPU.hpp:293-296)
        const double k0 = k2 * r0, k1 = k2 * r1;
#pragma unroll
        for (int d = 0; d < 3; d++) { U0[d] += k0 * jer[d]; U1[d] += k1 * jer[d]; }
        wH -= wB * sqrMagBdr * invF * invF;
        P += pc.chi[3] * (v2 * violaBdr);
        needReverse = true;
    }
#pragma unroll
    for (int d = 0; d < 3; d++) a2[d] = 2.0 * wH * h[d];

    if (needReverse) {
        // W = dxB^T U0 + dyB^T U1 + dzB^T U2 without forming the Jacobians (header comment)
        double zxU0[3], U0xy[3];
        cross3(zB, U0, zxU0);
        cross3(U0, yB, U0xy);
        const double Pv[3] = {U1[0] + zxU0[0], U1[1] + zxU0[1], U1[2] + zxU0[2]};
        const double yP = dot3(yB, Pv);
        const double Pp1 = (Pv[1] - yB[1] * yP) * invM, Pp2 = (Pv[2] - yB[2] * yP) * invM;   // G(czB) Pv, rows y,z
        const double q[3] = {U2[0] + U0xy[0], U2[1] + U0xy[1] - Pp2, U2[2] + U0xy[2] + Pp1};
        const double zq = dot3(zB, q);
#pragma unroll
        for (int d = 0; d < 3; d++) a2[d] += (q[d] - zB[d] * zq) * invF;
    }

    Psum = P;
    gTalpha = dot3(a0, vel) + dot3(a1, acc) + dot3(a2, jer) + dot3(a3, sna);
#pragma unroll
    for (int d = 0; d < 3; d++) { adj[d] = a0[d]; adj[3 + d] = a1[d]; adj[6 + d] = a2[d]; adj[9 + d] = a3[d]; }
}

// C2 / exponential time diffeomorphism, forward and derivative (CPU.hpp:639-641, 826-839)
FRX_HD double tau_to_T(double t, bool c2) {
    if (!c2) return exp(t);
    return t > 0.0 ? ((0.5 * t + 1.0) * t + 1.0) : 1.0 / ((0.5 * t - 1.0) * t + 1.0);
}
FRX_HD double dT_dtau(double t, bool c2) {
    if (!c2) return exp(t);
    if (t > 0.0) return t + 1.0;
    const double den = (0.5 * t - 1.0) * t + 1.0;
    return (1.0 - t) / (den * den);
}

} // namespace frx
