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
// phase boundary for the instruction scheduler (device only): keeps the register-hungry phases of a sample from
// being interleaved for ILP, which costs occupancy on a kernel that is FP64-issue bound anyway
#if defined(__HIP_DEVICE_COMPILE__)
#define FRX_PHASE() __builtin_amdgcn_sched_barrier(0)
#define FRX_LDS_READS_STAY_ABOVE() __builtin_amdgcn_sched_barrier(0x1)   // arithmetic may cross, memory operations may not: LDS reads issued above stay in flight together
#else
#define FRX_PHASE() ((void)0)
#define FRX_LDS_READS_STAY_ABOVE() ((void)0)
#endif

#ifndef FRX_HS_CHUNK
#define FRX_HS_CHUNK 4         // half-spaces whose sign tests run interleaved (penalty_sample); 8 needs 64 registers for the records alone and spills
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

// W = dxB^T U0 + dyB^T U1 + dzB^T U2 for the flat-output frame (xB, yB, zB)(h) without forming the Jacobians:
//   G(h)((0, -P'z, P'y) + U2 + U0 x yB),  P' = G(czB)(U1 + zB x U0),  G(x) v = (v - x^ (x^.v)) / |x|   (file header)
FRX_HD void frame_reverse(const double *U0, const double *U1, const double *U2, const double *zB, const double *yB, double invF,
                          double invM, double *W) {
    double zxU0[3], U0xy[3];
    cross3(zB, U0, zxU0);
    U0xy[0] = U0[1] * yB[2] - U0[2] * yB[1]; U0xy[1] = -(U0[0] * yB[2]); U0xy[2] = U0[0] * yB[1];   // U0 x yB, yB[0] = 0
    const double Pv[3] = {U1[0] + zxU0[0], U1[1] + zxU0[1], U1[2] + zxU0[2]};
    const double yP = yB[1] * Pv[1] + yB[2] * Pv[2];
    const double Pp1 = (Pv[1] - yB[1] * yP) * invM, Pp2 = (Pv[2] - yB[2] * yP) * invM;
    const double q[3] = {U2[0] + U0xy[0], U2[1] + U0xy[1] - Pp2, U2[2] + U0xy[2] + Pp1};
    const double zq = dot3(zB, q);
    W[0] = (q[0] - zB[0] * zq) * invF; W[1] = (q[1] - zB[1] * zq) * invF; W[2] = (q[2] - zB[2] * zq) * invF;
}

// Evaluate derivative `D` (0..4) of the quintic c[k*3+d] at s1 for the three axes.
template <int D, class CP> FRX_HD void poly_eval(CP c, double s1, double *out) {
    const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const double c1 = c[3 + d], c2 = c[6 + d], c3 = c[9 + d], c4 = c[12 + d], c5 = c[15 + d];
        if (D == 0) out[d] = c[d] + c1 * s1 + c2 * s2 + c3 * s3 + c4 * s4 + c5 * (s4 * s1);
        if (D == 1) out[d] = c1 + c2 * (2.0 * s1) + c3 * (3.0 * s2) + c4 * (4.0 * s3) + c5 * (5.0 * s4);
        if (D == 2) out[d] = c2 * 2.0 + c3 * (6.0 * s1) + c4 * (12.0 * s2) + c5 * (20.0 * s3);
        if (D == 3) out[d] = c3 * 6.0 + c4 * (24.0 * s1) + c5 * (60.0 * s2);
        if (D == 4) out[d] = c4 * 24.0 + c5 * (120.0 * s1);
    }
}

// 1/sqrt(x).  Device: v_rsq_f64 (about 2^-23 relative) refined by one third-order and one second-order step - 9 instructions
// instead of the ~26 of an IEEE sqrt followed by an IEEE division; relative error < 2^-51 (not correctly rounded: the reference's
// 1/sqrt differs from it in the last bit at most, far inside the parity tolerances).  Host: the plain expression.
FRX_HD double rsqrt_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-x * y, y, 1.0);
    y = __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);
    e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(y * e, 0.5, y);
#else
    return 1.0 / sqrt(x);
#endif
}

// Read-only view of doubles behind a pointer; fence() is where a device view forgets what it has read (see LdsView in
// frx_kernels.hpp).  The host and the generic device path use this one.
struct PtrView {
    const double *p;
    FRX_HD double operator[](int i) const { return p[i]; }
    FRX_HD void fence() {}
};

// One quadrature sample of piece coefficients c[k*3+d] (k = power, d = axis) at local time s1.
//   c     = view of the 18 coefficients.  The sample is evaluated in PHASES (attitude -> half-space loop -> one derivative at a
//           time) and c.fence() is called between them: on the device the view is an LDS address that is made opaque at every
//           fence, so vel/acc/jer/sna and everything derived from them are evaluated AFTER the half-space loop from fresh
//           ds_read_b64 pairs instead of being kept live across it (occupancy).  (Until round 2 the same effect came from a
//           volatile-qualified generic pointer, which compiled to 60 serialised flat loads, each followed by s_waitcnt vmcnt(0).)
//   ws    = omega * step  (trapezoid weight x step, CPU.hpp:245,306)
//   hb    = view of the piece's corridor block: {origin xyz, K} then K records of 4 doubles (n_x, n_y, n_z, c) with unit normal
//           and c = n.(p_k - org) - safeMargin, so that  n.(pos - p_k) + safeMargin = n.(pos - org) - c.
//           The origin (the point of the polytope's first half-space) is subtracted once per sample and keeps the magnitudes at
//           corridor scale (metres), as pos - p_k does in CPU.hpp:325.
//   K, Kub = the piece's number of half-spaces and a wave-uniform bound on it (records k < Kub are readable: the block is zero-padded)
//   adj   = out: a0,a1,a2,a3 (12 doubles), already weighted by ws
//   Psum  = out: sum of chi*viol^3 (not weighted)
//   gTalpha = out: a0.v + a1.a + a2.j + a3.s  (to be multiplied by alpha = j/kappa)
//   LAT   = latency form: no phase boundaries and no forgetting - the scheduler may interleave the corridor loop with the limits (which
//           do not depend on it) at the price of registers; for launches that cannot fill the chip anyway (one wave's latency IS the launch)
template <bool LAT, class CV, class HV>
FRX_HD void penalty_sample(CV c, double s1, double ws, const PenaltyConst &pc, HV hb, int K, int Kub, double *adj, double &Psum, double &gTalpha) {
    // ---- phase A: position and attitude only (CPU.hpp:260-279); everything else is evaluated after the loop ----
    double pl[3], zB[3], yB[3], xB[3], invF, invM;
    {
        double pos[3], acc[3];
        poly_eval<0>(c, s1, pos);
        poly_eval<2>(c, s1, acc);
        const double h[3] = {acc[0], acc[1], acc[2] + pc.gAcc};
        invF = rsqrt_fast(dot3(h, h));
#pragma unroll
        for (int d = 0; d < 3; d++) { zB[d] = h[d] * invF; pl[d] = pos[d] - hb[d]; }
        invM = rsqrt_fast(zB[2] * zB[2] + zB[1] * zB[1]);
        yB[0] = 0.0; yB[1] = zB[2] * invM; yB[2] = -zB[1] * invM;
        // xB = yB x zB with yB[0] = 0 written out (cross3 multiplied by the literal zero twice: 0 * x may not be folded)
        xB[0] = yB[1] * zB[2] - yB[2] * zB[1]; xB[1] = yB[2] * zB[0]; xB[2] = -(yB[1] * zB[0]);
    }
    if (!LAT) { c.fence(); FRX_PHASE(); }

    double a0[3] = {0, 0, 0};
    double U0[3] = {0, 0, 0}, U1[3] = {0, 0, 0}, U2[3] = {0, 0, 0};
    double P = 0.0;
    bool needReverse = false;

    // ---- corridor half-spaces (CPU.hpp:310-345); the sign test avoids the sqrt unless violated ----
    // Two passes over chunks of FRX_HS_CHUNK half-spaces (round 4).  Pass 1 runs the sign test of the whole chunk WITHOUT a branch: independent
    // chains that the scheduler interleaves, their records read from LDS in one go - the one-loop form tested, branched and read half-space by
    // half-space, so a lone wave paid an LDS latency plus a dependent chain per half-space (ISA: ds_read2 / s_waitcnt / 18 VALU / s_cbranch, eight
    // times: ~1.8 k of a sample's ~5 k cycles).  Pass 2 revisits, in the same order, only the half-spaces pass 1 flagged (none for a sample inside
    // its corridor) and does what the branch did.  Same expressions, same order of accumulation.
    // Round 5 - pre-reject: the ellipsoid's support in direction n is  |E R^T n| <= max(ell) |R^T n| = max(ell)  (R orthonormal, n a unit normal),
    // so  d0 < -max(ell)  proves  d0 < 0  and  |E R^T n|^2 <= max(ell)^2 < d0^2 : the half-space cannot be flagged, and the three frame
    // projections, their squares and the comparison against d0^2 (18 of the 22 FP64 instructions of a test) are never issued for it.  The guard
    // factor (1 + 2^-20) covers the rounding of the computed |R^T n|^2 (a few ulp) many times over; whatever is not rejected takes the full test
    // of round 4, so the flags - and with them every result bit - are the ones the full test alone would give.  Measured on the headline batch
    // (scripts/r05/prereject_stats.py, CPU): 98.9 % of the (sample, half-space) pairs and 90 % of the (wave, chunk) pairs are rejected at the
    // converged trajectories, 98 % of the (wave, chunk) pairs along the way.
    // Kub: wave-uniform bound on K (the corridor block is zero-padded to it); lanes with fewer half-spaces mask the surplus.
    {
        const double e0 = pc.ell[0], e1 = pc.ell[1], e2 = pc.ell[2];
        const double emax = (e0 > e1 ? (e0 > e2 ? e0 : e2) : (e1 > e2 ? e1 : e2)) * (1.0 + 9.5367431640625e-7);
        double Pcorr = 0.0;
        constexpr int CH = FRX_HS_CHUNK;
        for (int k0 = 0; k0 < Kub; k0 += CH) {
            unsigned need = 0u;
            unsigned long long maybe = 0;
            double rec[CH][4];
            FRX_PHASE();                                                   // between this fence and the next: the chunk's LDS reads and nothing else
#if defined(FRX_COUNT_BUILD)                                     // scripts/count_fp64.py: one copy of the test, so that static counts are per half-space
#pragma unroll 1
#else
#pragma unroll
#endif
            for (int j = 0; j < CH; j++) {                                 // the chunk's records first, all reads in flight together ...
#if defined(__HIP_DEVICE_COMPILE__)
                const int r = 4 + 4 * (k0 + j);                            // device: up to CH - 1 records past the piece's block are still the wave's own LDS (the next
#else                                                                      // block or the reduction slots behind them); what is read there is masked by k < K below
                const int k = k0 + j, r = 4 + 4 * (k < Kub ? k : Kub - 1);
#endif
                rec[j][0] = hb[r]; rec[j][1] = hb[r + 1]; rec[j][2] = hb[r + 2]; rec[j][3] = hb[r + 3];
            }
            FRX_PHASE();                                                   // (... not re-sunk next to their uses: the scheduler minimises registers, not latency)
#if !defined(FRX_NO_PREREJECT)
#if defined(FRX_COUNT_BUILD)
#pragma unroll 1
#else
#pragma unroll
#endif
            for (int j = 0; j < CH; j++) {                                 // pre-reject: the signed distance of the CENTRE alone
                const double *n = rec[j];
                const double d0 = dot3(n, pl) - n[3];                      // n.(pos - p_k) + safeMargin
                const bool open_j = (k0 + j < K) & !(d0 < -emax);          // (NaN stays "open": the full test decides)
#if defined(__HIP_DEVICE_COMPILE__)
                // ONE wave-uniform flag for the chunk (lane masks combined on the scalar unit, a scalar branch): a lane whose own half-spaces are all
                // rejected runs the full test with its wave when another lane needs it - the full test is the authority, its flags are the same
                maybe |= __builtin_amdgcn_ballot_w64(open_j);
#else
                maybe |= open_j ? 1u : 0u;
#endif
            }
            if (maybe != 0) {
#else
            maybe = 1;
            {
#endif
#if defined(FRX_COUNT_BUILD)
#pragma unroll 1
#else
#pragma unroll
#endif
                for (int j = 0; j < CH; j++) {
                    const double *n = rec[j];
                    // (R^T n) .* ellipsoid
                    const double w0 = dot3(xB, n) * e0, w1 = (yB[1] * n[1] + yB[2] * n[2]) * e1, w2 = dot3(zB, n) * e2;
                    const double eN2 = w0 * w0 + w1 * w1 + w2 * w2;
                    const double d0 = dot3(n, pl) - n[3];                  // n.(pos - p_k) + safeMargin
                    need |= (k0 + j < K && (d0 >= 0.0 || eN2 > d0 * d0)) ? (1u << j) : 0u;
                }
            }
            if (need != 0u) {
#pragma unroll 1
                for (int j = 0; j < CH; j++) {
                    if (!(need & (1u << j))) continue;
                    const int r = 4 + 4 * (k0 + j);
                    const double n[3] = {hb[r], hb[r + 1], hb[r + 2]};
                    const double w0 = dot3(xB, n) * e0, w1 = (yB[1] * n[1] + yB[2] * n[2]) * e1, w2 = dot3(zB, n) * e2;
                    const double eN2 = w0 * w0 + w1 * w1 + w2 * w2;
                    const double d0 = dot3(n, pl) - hb[r + 3];
                    const double ir = rsqrt_fast(eN2), eNorm = eN2 * ir;
                    const double sd = d0 + eNorm;                          // CPU.hpp:325,328
                    if (sd > 0.0) {
                        const double sd2 = sd * sd;
                        const double cw = ws * pc.chi[0] * 3.0 * sd2;
                        const double ie = cw * ir;
                        const double cg0 = w0 * ie * e0, cg1 = w1 * ie * e1, cg2 = w2 * ie * e2;   // cw * eNormGd, CPU.hpp:324,326
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
        }
        P += pc.chi[0] * Pcorr;
    }
    if (!LAT) FRX_PHASE();

    // ---- phase B: limits (CPU.hpp:281-304, 347-398) and reverse passes, one derivative at a time so that
    //      vel/acc/jer/sna are never live together; gTalpha = a0.v + a1.a + a2.j + a3.s is accumulated on the way ----
    double a2[3] = {0, 0, 0};
    if (needReverse) frame_reverse(U0, U1, U2, zB, yB, invF, invM, a2);      // corridor part of d pen / d acc
    if (!LAT) FRX_PHASE();
    double gT;
    {   // velocity limit (CPU.hpp:347-359)
        double vel[3];
        poly_eval<1>(c, s1, vel);
        gT = dot3(a0, vel);
        const double violaVel = dot3(vel, vel) - pc.vMaxSqr;
        double wV = 0.0;
        if (violaVel > 0.0) {
            const double v2 = violaVel * violaVel;
            wV = ws * pc.chi[1] * 3.0 * v2 * 2.0;
            P += pc.chi[1] * (v2 * violaVel);
        }
#pragma unroll
        for (int d = 0; d < 3; d++) adj[3 + d] = wV * vel[d];
    }
    if (!LAT) { c.fence(); FRX_PHASE(); }
    double h[3], wH = 0.0;                                      // weight on dSqrMagThr = 2h
    {   // thrust limits (CPU.hpp:361-385; both use chi[2])
        double acc[3];
        poly_eval<2>(c, s1, acc);
        gT += adj[3] * acc[0] + adj[4] * acc[1] + adj[5] * acc[2];
        h[0] = acc[0]; h[1] = acc[1]; h[2] = acc[2] + pc.gAcc;
        const double sqrMagThr = dot3(h, h);                   // the reference squares the square root of this (CPU.hpp:285-287): one rounding apart
        const double violaThrl = pc.thrMinSqr - sqrMagThr, violaThrh = sqrMagThr - pc.thrMaxSqr;
        if (violaThrl > 0.0) {
            const double v2 = violaThrl * violaThrl;
            wH -= ws * pc.chi[2] * 3.0 * v2;
            P += pc.chi[2] * (v2 * violaThrl);
        }
        if (violaThrh > 0.0) {
            const double v2 = violaThrh * violaThrh;
            wH += ws * pc.chi[2] * 3.0 * v2;
            P += pc.chi[2] * (v2 * violaThrh);
        }
    }
    if (!LAT) { c.fence(); FRX_PHASE(); }
    double a3[3] = {0, 0, 0};
    {   // body-rate limit (CPU.hpp:285-299, 387-398)
        double jer[3];
        poly_eval<3>(c, s1, jer);
        const double r0 = dot3(xB, jer), r1 = yB[1] * jer[1] + yB[2] * jer[2];   // (yB[0] = 0: the compiler may not drop a 0 * x)
        const double b0 = r0 * invF, b1 = r1 * invF;
        const double sqrMagBdr = b1 * b1 + b0 * b0;
        const double violaBdr = sqrMagBdr - pc.bdrMaxSqr;
        if (violaBdr > 0.0) {
            const double v2 = violaBdr * violaBdr;
            const double wB = ws * pc.chi[3] * 3.0 * v2;
            const double k2 = wB * 2.0 * invF * invF;
            // d(omega^2)/d jer = (2/f^2)(r0 xB + r1 yB)   (= dJerSqrMagBdr, CPU.hpp:297-299)
#pragma unroll
            for (int d = 0; d < 3; d++) a3[d] = k2 * (r0 * xB[d] + r1 * yB[d]);
            // d(omega^2)/d h = (2/f^2)(r0 dxB^T jer + r1 dyB^T jer) - 2 omega^2 h / f^2   (= dSqrMagBdr, CPU.hpp:293-296)
            const double k0 = k2 * r0, k1 = k2 * r1;
            const double V0[3] = {k0 * jer[0], k0 * jer[1], k0 * jer[2]}, V1[3] = {k1 * jer[0], k1 * jer[1], k1 * jer[2]};
            const double V2[3] = {0.0, 0.0, 0.0};
            double Wb[3];
            frame_reverse(V0, V1, V2, zB, yB, invF, invM, Wb);
#pragma unroll
            for (int d = 0; d < 3; d++) a2[d] += Wb[d];
            wH -= wB * sqrMagBdr * invF * invF;
            P += pc.chi[3] * (v2 * violaBdr);
        }
#pragma unroll
        for (int d = 0; d < 3; d++) a2[d] += 2.0 * wH * h[d];
        gT += dot3(a2, jer);
    }
    if (!LAT) { c.fence(); FRX_PHASE(); }
    {
        double sna[3];
        poly_eval<4>(c, s1, sna);
        gT += dot3(a3, sna);
    }
    Psum = P;
    gTalpha = gT;
#pragma unroll
    for (int d = 0; d < 3; d++) { adj[d] = a0[d]; adj[6 + d] = a2[d]; adj[9 + d] = a3[d]; }
}
// Pointer form (host check, tests/hostcheck): coefficients c, origin org (3 doubles), K records hs (4 doubles each).
FRX_HD void penalty_sample(const double *c, const double *, double s1, double ws, const PenaltyConst &pc, const double *org,
                           const double *hs, int K, double *adj, double &Psum, double &gTalpha) {
    struct Blk {                       // {origin xyz, K} + records, addressed like the device's corridor block
        const double *org, *hs;
        FRX_HD double operator[](int i) const { return i < 4 ? org[i < 3 ? i : 0] : hs[i - 4]; }
    };
    penalty_sample<false>(PtrView{c}, s1, ws, pc, Blk{org, hs}, K, K, adj, Psum, gTalpha);
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
