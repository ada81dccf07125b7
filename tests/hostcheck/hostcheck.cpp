// TEST INFRASTRUCTURE — compiles the product's per-sample math header (fast-racing_amd/csrc/frx_math.hpp)
// for the HOST so its algebra (reverse-mode adjoints instead of the reference's explicit Jacobians) can be
// checked against the oracle on a CPU-only box.  Not part of the product library; never used as a fallback.
#include "../../fast-racing_amd/csrc/frx_math.hpp"

extern "C" void hostcheck_penalty(int N, int kappa, const double *T, const double *C, const int *hoff, const double *hrec,
                                  const double *pc9_chi4 /* ell[3], margin, vMax, thrMin, thrMax, bdrMax, g, chi[4] */,
                                  int abscissa_accumulate, double *out20 /* N x 20: cost, gdT, gdC[18] */) {
    frx::PenaltyConst pc;
    pc.ell[0] = pc9_chi4[0]; pc.ell[1] = pc9_chi4[1]; pc.ell[2] = pc9_chi4[2];
    pc.safeMargin = pc9_chi4[3];
    pc.vMaxSqr = pc9_chi4[4] * pc9_chi4[4]; pc.thrMinSqr = pc9_chi4[5] * pc9_chi4[5];
    pc.thrMaxSqr = pc9_chi4[6] * pc9_chi4[6]; pc.bdrMaxSqr = pc9_chi4[7] * pc9_chi4[7];
    pc.gAcc = pc9_chi4[8];
    for (int q = 0; q < 4; q++) pc.chi[q] = pc9_chi4[9 + q];
    for (int i = 0; i < N; i++) {
        const double *c = C + 18 * i;
        double *o = out20 + 20 * i;
        for (int v = 0; v < 20; v++) o[v] = 0.0;
        const double step = T[i] / kappa;
        double s1acc = 0.0;
        for (int j = 0; j <= kappa; j++) {
            const double s1 = abscissa_accumulate ? s1acc : step * j;
            const double omg = (j == 0 || j == kappa) ? 0.5 : 1.0;
            const double alpha = 1.0 / kappa * j;
            double adj[12], P, gTa;
            frx::penalty_sample(c, s1, omg * step, pc, hrec + 6 * hoff[i], hoff[i + 1] - hoff[i], adj, P, gTa);
            const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
            const double b0[6] = {1.0, s1, s2, s3, s4, s5};
            const double b1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
            const double b2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
            const double b3[6] = {0.0, 0.0, 0.0, 6.0, 24.0 * s1, 60.0 * s2};
            o[0] += omg * step * P;
            o[1] += alpha * gTa + omg * P / kappa;
            for (int k = 0; k < 6; k++)
                for (int d = 0; d < 3; d++)
                    o[2 + 3 * k + d] += b0[k] * adj[d] + b1[k] * adj[3 + d] + b2[k] * adj[6 + d] + b3[k] * adj[9 + d];
            s1acc += step;
        }
    }
}
