// TEST INFRASTRUCTURE — CPU oracle for the SE(3) MINCO cost/gradient hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (fast-racing_amd/csrc) never links, includes or calls anything in oracle/.
//
// PARITY PIN: the reference holds no tests, golden vectors or fixtures for plan_manage
// (SURVEY.md §4, §8c).  This file is a plain-C++ restatement (no Eigen) that follows the
// reference line by line; each function cites the lines it restates.  It is PINNED against
// outputs of the reference itself run in this container:
//  (a) oracle/_ref/libref_gcopter.so = the reference's CPU path (se3gcopter_cpu.hpp, trajectory.hpp,
//      geoutils.hpp, sdlp.hpp, quickhull.hpp, lbfgs.hpp) compiled UNMODIFIED where it lies, against
//      oracle/eigen_shim (a minimal stand-in for the Eigen API those headers use — Eigen is not
//      installed); tests/test_reference_pin.py: initial guess, forward map, penalty integrator and the
//      full L-BFGS callback agree to 1e-15 ... 1e-10; the committed fixtures tests/golden/*.npz carry
//      that library's outputs (ref_* arrays) to the GPU box;
//  (b) oracle/_ref/libref_lbfgs.so = the reference's lbfgs.hpp alone: iterates bit-identical
//      (tests/test_lbfgs.py);
//  (c) the self-checks of SURVEY.md §8c (finite differences, dense linear algebra, spline
//      invariants) in tests/test_oracle.py.
//
// reference files (relative to /root/reference/src/plan_manage/include/se3gcopter/):
//   CPU.hpp  = se3gcopter_cpu.hpp      traj.hpp = trajectory.hpp      lbfgs.hpp
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "lbfgs_port.hpp"

namespace orc {

// ------------------------------------------------------------------------------------------
// BandedSystem — traj.hpp:599-752.  Band storage ptrData[(i-j+upperBw)*N + j] (traj.hpp:645).
// ------------------------------------------------------------------------------------------
struct Banded {
    int N = 0, lowerBw = 0, upperBw = 0;
    std::vector<double> d;
    void create(int n, int p, int q) { N = n; lowerBw = p; upperBw = q; d.assign((size_t)N * (p + q + 1), 0.0); }
    void reset() { std::fill(d.begin(), d.end(), 0.0); }
    inline double &at(int i, int j) { return d[(size_t)(i - j + upperBw) * N + j]; }
    inline const double &at(int i, int j) const { return d[(size_t)(i - j + upperBw) * N + j]; }

    // traj.hpp:655-687 — in-place LU, NO pivoting, exact zeros skipped
    void factorizeLU() {
        for (int k = 0; k <= N - 2; k++) {
            int iM = std::min(k + lowerBw, N - 1);
            double cVl = at(k, k);
            for (int i = k + 1; i <= iM; i++)
                if (at(i, k) != 0.0) at(i, k) /= cVl;
            int jM = std::min(k + upperBw, N - 1);
            for (int j = k + 1; j <= jM; j++) {
                cVl = at(k, j);
                if (cVl != 0.0)
                    for (int i = k + 1; i <= iM; i++)
                        if (at(i, k) != 0.0) at(i, j) -= at(i, k) * cVl;
            }
        }
    }
    // traj.hpp:692-719 — b is N x m, row-major here (the reference's b.row(i) operations)
    void solve(double *b, int m) const {
        for (int j = 0; j <= N - 1; j++) {
            int iM = std::min(j + lowerBw, N - 1);
            for (int i = j + 1; i <= iM; i++)
                if (at(i, j) != 0.0)
                    for (int c = 0; c < m; c++) b[i * m + c] -= at(i, j) * b[j * m + c];
        }
        for (int j = N - 1; j >= 0; j--) {
            for (int c = 0; c < m; c++) b[j * m + c] /= at(j, j);
            int iM = std::max(0, j - upperBw);
            for (int i = iM; i <= j - 1; i++)
                if (at(i, j) != 0.0)
                    for (int c = 0; c < m; c++) b[i * m + c] -= at(i, j) * b[j * m + c];
        }
    }
    // traj.hpp:724-751 — solves A^T x = b with the same factors
    void solveAdj(double *b, int m) const {
        for (int j = 0; j <= N - 1; j++) {
            for (int c = 0; c < m; c++) b[j * m + c] /= at(j, j);
            int iM = std::min(j + upperBw, N - 1);
            for (int i = j + 1; i <= iM; i++)
                if (at(j, i) != 0.0)
                    for (int c = 0; c < m; c++) b[i * m + c] -= at(j, i) * b[j * m + c];
        }
        for (int j = N - 1; j >= 0; j--) {
            int iM = std::max(0, j - lowerBw);
            for (int i = iM; i <= j - 1; i++)
                if (at(j, i) != 0.0)
                    for (int c = 0; c < m; c++) b[i * m + c] -= at(j, i) * b[j * m + c];
        }
    }
};

// small fixed-size helpers (3-vectors, 3x3 row-major matrices M[r][c])
struct V3 { double v[3]; double &operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
struct M3 { double m[3][3]; };
static inline double dot3(const V3 &a, const V3 &b) { return a(0) * b(0) + a(1) * b(1) + a(2) * b(2); }
static inline V3 cross3(const V3 &a, const V3 &b) {
    return V3{{a(1) * b(2) - a(2) * b(1), a(2) * b(0) - a(0) * b(2), a(0) * b(1) - a(1) * b(0)}};
}
static inline V3 col(const M3 &A, int c) { return V3{{A.m[0][c], A.m[1][c], A.m[2][c]}}; }
static inline V3 matvec(const M3 &A, const V3 &x) {
    V3 r;
    for (int i = 0; i < 3; i++) r(i) = A.m[i][0] * x(0) + A.m[i][1] * x(1) + A.m[i][2] * x(2);
    return r;
}
static inline V3 matTvec(const M3 &A, const V3 &x) {   // A^T x
    V3 r;
    for (int i = 0; i < 3; i++) r(i) = A.m[0][i] * x(0) + A.m[1][i] * x(1) + A.m[2][i] * x(2);
    return r;
}
static inline M3 matmul(const M3 &A, const M3 &B) {
    M3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}

// CPU.hpp:163-185
static inline void normalizeFDF(const V3 &x, V3 &xNor, M3 &G) {
    const double a = x(0), b = x(1), c = x(2);
    const double aSqr = a * a, bSqr = b * b, cSqr = c * c;
    const double ab = a * b, bc = b * c, ca = c * a;
    const double xSqrNorm = aSqr + bSqr + cSqr;
    const double xNorm = std::sqrt(xSqrNorm);
    const double den = xSqrNorm * xNorm;
    xNor = V3{{a / xNorm, b / xNorm, c / xNorm}};
    G.m[0][0] = bSqr + cSqr; G.m[0][1] = -ab;         G.m[0][2] = -ca;
    G.m[1][0] = -ab;         G.m[1][1] = aSqr + cSqr; G.m[1][2] = -bc;
    G.m[2][0] = -ca;         G.m[2][1] = -bc;         G.m[2][2] = aSqr + bSqr;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) G.m[i][j] /= den;
}

struct PenaltyParams {           // the argument list of addTimeIntPenalty, CPU.hpp:188-201
    double ellipsoid[3];
    double safeMargin, vMax, thrAccMin, thrAccMax, bdrMax, gAcc;
    double ci[4];
};

// ------------------------------------------------------------------------------------------
// MINCO_S3 — CPU.hpp:37-565.  b / gdC are (6N x 3) row-major: row 6i+k = coefficient of t^k of
// piece i (CPU.hpp:244,260).
// ------------------------------------------------------------------------------------------
struct MincoS3 {
    int N = 0;
    double headPVA[9], tailPVA[9];      // column-major 3x3: col0 = p, col1 = v, col2 = a (CPU.hpp:440-442)
    std::vector<double> T1, T2, T3, T4, T5;
    Banded A;
    std::vector<double> b, gdC;
    bool s1_accumulate = true;          // true: s1 += step (CPU.hpp:246,400); false: s1 = step*j (cuda_computer.cu:152)

    inline const double *row(int r) const { return &b[(size_t)r * 3]; }
    static inline double sq3(const double *a) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }
    static inline double dt3(const double *a, const double *c) { return a[0] * c[0] + a[1] * c[1] + a[2] * c[2]; }

    // CPU.hpp:411-423
    void reset(const double *headState, const double *tailState, int pieceNum) {
        N = pieceNum;
        std::memcpy(headPVA, headState, sizeof(headPVA));
        std::memcpy(tailPVA, tailState, sizeof(tailPVA));
        T1.assign(N, 0.0);
        A.create(6 * N, 6, 6);
        b.assign((size_t)6 * N * 3, 0.0);
        gdC.assign((size_t)6 * N * 3, 0.0);
    }

    // CPU.hpp:425-505
    void generate(const double *inPs /*3 x (N-1) col-major*/, const double *ts) {
        T1.assign(ts, ts + N);
        T2.resize(N); T3.resize(N); T4.resize(N); T5.resize(N);
        for (int i = 0; i < N; i++) {
            T2[i] = T1[i] * T1[i];
            T3[i] = T2[i] * T1[i];
            T4[i] = T2[i] * T2[i];
            T5[i] = T4[i] * T1[i];
        }
        A.reset();
        std::fill(b.begin(), b.end(), 0.0);

        A.at(0, 0) = 1.0; A.at(1, 1) = 1.0; A.at(2, 2) = 2.0;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) b[r * 3 + c] = headPVA[r * 3 + c];   // b.row(r) = headPVA.col(r)^T

        for (int i = 0; i < N - 1; i++) {
            A.at(6 * i + 3, 6 * i + 3) = 6.0;
            A.at(6 * i + 3, 6 * i + 4) = 24.0 * T1[i];
            A.at(6 * i + 3, 6 * i + 5) = 60.0 * T2[i];
            A.at(6 * i + 3, 6 * i + 9) = -6.0;
            A.at(6 * i + 4, 6 * i + 4) = 24.0;
            A.at(6 * i + 4, 6 * i + 5) = 120.0 * T1[i];
            A.at(6 * i + 4, 6 * i + 10) = -24.0;
            A.at(6 * i + 5, 6 * i) = 1.0;
            A.at(6 * i + 5, 6 * i + 1) = T1[i];
            A.at(6 * i + 5, 6 * i + 2) = T2[i];
            A.at(6 * i + 5, 6 * i + 3) = T3[i];
            A.at(6 * i + 5, 6 * i + 4) = T4[i];
            A.at(6 * i + 5, 6 * i + 5) = T5[i];
            A.at(6 * i + 6, 6 * i) = 1.0;
            A.at(6 * i + 6, 6 * i + 1) = T1[i];
            A.at(6 * i + 6, 6 * i + 2) = T2[i];
            A.at(6 * i + 6, 6 * i + 3) = T3[i];
            A.at(6 * i + 6, 6 * i + 4) = T4[i];
            A.at(6 * i + 6, 6 * i + 5) = T5[i];
            A.at(6 * i + 6, 6 * i + 6) = -1.0;
            A.at(6 * i + 7, 6 * i + 1) = 1.0;
            A.at(6 * i + 7, 6 * i + 2) = 2 * T1[i];
            A.at(6 * i + 7, 6 * i + 3) = 3 * T2[i];
            A.at(6 * i + 7, 6 * i + 4) = 4 * T3[i];
            A.at(6 * i + 7, 6 * i + 5) = 5 * T4[i];
            A.at(6 * i + 7, 6 * i + 7) = -1.0;
            A.at(6 * i + 8, 6 * i + 2) = 2.0;
            A.at(6 * i + 8, 6 * i + 3) = 6 * T1[i];
            A.at(6 * i + 8, 6 * i + 4) = 12 * T2[i];
            A.at(6 * i + 8, 6 * i + 5) = 20 * T3[i];
            A.at(6 * i + 8, 6 * i + 8) = -2.0;
            for (int c = 0; c < 3; c++) b[(6 * i + 5) * 3 + c] = inPs[i * 3 + c];
        }
        A.at(6 * N - 3, 6 * N - 6) = 1.0;
        A.at(6 * N - 3, 6 * N - 5) = T1[N - 1];
        A.at(6 * N - 3, 6 * N - 4) = T2[N - 1];
        A.at(6 * N - 3, 6 * N - 3) = T3[N - 1];
        A.at(6 * N - 3, 6 * N - 2) = T4[N - 1];
        A.at(6 * N - 3, 6 * N - 1) = T5[N - 1];
        A.at(6 * N - 2, 6 * N - 5) = 1.0;
        A.at(6 * N - 2, 6 * N - 4) = 2 * T1[N - 1];
        A.at(6 * N - 2, 6 * N - 3) = 3 * T2[N - 1];
        A.at(6 * N - 2, 6 * N - 2) = 4 * T3[N - 1];
        A.at(6 * N - 2, 6 * N - 1) = 5 * T4[N - 1];
        A.at(6 * N - 1, 6 * N - 4) = 2;
        A.at(6 * N - 1, 6 * N - 3) = 6 * T1[N - 1];
        A.at(6 * N - 1, 6 * N - 2) = 12 * T2[N - 1];
        A.at(6 * N - 1, 6 * N - 1) = 20 * T3[N - 1];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) b[(6 * N - 3 + r) * 3 + c] = tailPVA[r * 3 + c];

        A.factorizeLU();
        A.solve(b.data(), 3);
    }

    // CPU.hpp:507-520
    double getTrajJerkCost() const {
        double objective = 0.0;
        for (int i = 0; i < N; i++) {
            const double *c3 = row(6 * i + 3), *c4 = row(6 * i + 4), *c5 = row(6 * i + 5);
            objective += 36.0 * sq3(c3) * T1[i] +
                         144.0 * dt3(c4, c3) * T2[i] +
                         192.0 * sq3(c4) * T3[i] +
                         240.0 * dt3(c5, c3) * T3[i] +
                         720.0 * dt3(c5, c4) * T4[i] +
                         720.0 * sq3(c5) * T5[i];
        }
        return objective;
    }
    // CPU.hpp:65-77
    void addGradJbyT(double *gdT) const {
        for (int i = 0; i < N; i++) {
            const double *c3 = row(6 * i + 3), *c4 = row(6 * i + 4), *c5 = row(6 * i + 5);
            gdT[i] += 36.0 * sq3(c3) +
                      288.0 * dt3(c4, c3) * T1[i] +
                      576.0 * sq3(c4) * T2[i] +
                      720.0 * dt3(c5, c3) * T2[i] +
                      2880.0 * dt3(c5, c4) * T3[i] +
                      3600.0 * sq3(c5) * T4[i];
        }
    }
    // CPU.hpp:80-95
    void addGradJbyC(double *g) const {
        for (int i = 0; i < N; i++) {
            const double *c3 = row(6 * i + 3), *c4 = row(6 * i + 4), *c5 = row(6 * i + 5);
            for (int c = 0; c < 3; c++) {
                g[(6 * i + 5) * 3 + c] += 240.0 * c3[c] * T3[i] + 720.0 * c4[c] * T4[i] + 1440.0 * c5[c] * T5[i];
                g[(6 * i + 4) * 3 + c] += 144.0 * c3[c] * T2[i] + 384.0 * c4[c] * T3[i] + 720.0 * c5[c] * T4[i];
                g[(6 * i + 3) * 3 + c] += 72.0 * c3[c] * T1[i] + 144.0 * c4[c] * T2[i] + 240.0 * c5[c] * T3[i];
            }
        }
    }
    // CPU.hpp:104-151
    void addPropCtoT(const double *adj, double *gdT) const {
        for (int i = 0; i < N - 1; i++) {
            double negVel[3], negAcc[3], negJer[3], negSnp[3], negCrk[3];
            for (int c = 0; c < 3; c++) {
                negVel[c] = -(row(i * 6 + 1)[c] + 2.0 * T1[i] * row(i * 6 + 2)[c] + 3.0 * T2[i] * row(i * 6 + 3)[c] +
                              4.0 * T3[i] * row(i * 6 + 4)[c] + 5.0 * T4[i] * row(i * 6 + 5)[c]);
                negAcc[c] = -(2.0 * row(i * 6 + 2)[c] + 6.0 * T1[i] * row(i * 6 + 3)[c] + 12.0 * T2[i] * row(i * 6 + 4)[c] +
                              20.0 * T3[i] * row(i * 6 + 5)[c]);
                negJer[c] = -(6.0 * row(i * 6 + 3)[c] + 24.0 * T1[i] * row(i * 6 + 4)[c] + 60.0 * T2[i] * row(i * 6 + 5)[c]);
                negSnp[c] = -(24.0 * row(i * 6 + 4)[c] + 120.0 * T1[i] * row(i * 6 + 5)[c]);
                negCrk[c] = -120.0 * row(i * 6 + 5)[c];
            }
            const double *B1[6] = {negSnp, negCrk, negVel, negVel, negAcc, negJer};   // CPU.hpp:128
            // (B1 .* adj.block<6,3>(6i+3,0)).sum(): Eigen reduces column-major (column by column)
            double s = 0.0;
            for (int c = 0; c < 3; c++)
                for (int r = 0; r < 6; r++) s += B1[r][c] * adj[(6 * i + 3 + r) * 3 + c];
            gdT[i] += s;
        }
        double negVel[3], negAcc[3], negJer[3];
        for (int c = 0; c < 3; c++) {
            negVel[c] = -(row(6 * N - 5)[c] + 2.0 * T1[N - 1] * row(6 * N - 4)[c] + 3.0 * T2[N - 1] * row(6 * N - 3)[c] +
                          4.0 * T3[N - 1] * row(6 * N - 2)[c] + 5.0 * T4[N - 1] * row(6 * N - 1)[c]);
            negAcc[c] = -(2.0 * row(6 * N - 4)[c] + 6.0 * T1[N - 1] * row(6 * N - 3)[c] + 12.0 * T2[N - 1] * row(6 * N - 2)[c] +
                          20.0 * T3[N - 1] * row(6 * N - 1)[c]);
            negJer[c] = -(6.0 * row(6 * N - 3)[c] + 24.0 * T1[N - 1] * row(6 * N - 2)[c] + 60.0 * T2[N - 1] * row(6 * N - 1)[c]);
        }
        const double *B2[3] = {negVel, negAcc, negJer};
        double s = 0.0;
        for (int c = 0; c < 3; c++)
            for (int r = 0; r < 3; r++) s += B2[r][c] * adj[(6 * N - 3 + r) * 3 + c];
        gdT[N - 1] += s;
    }
    // CPU.hpp:154-161 — gdInP 3 x (N-1) col-major
    void addPropCtoP(const double *adj, double *gdInP) const {
        for (int i = 0; i < N - 1; i++)
            for (int c = 0; c < 3; c++) gdInP[i * 3 + c] += adj[(6 * i + 5) * 3 + c];
    }

    // CPU.hpp:188-408 — the hot loop.  cfgHs[idx] is 6 x K column-major: column k = (n_k; p_k).
    void addTimeIntPenalty(const int *cons, const int *idxHs, const std::vector<std::vector<double>> &cfgHs,
                           const PenaltyParams &pp, double &cost, double *gdT, double *gdCacc) const {
        double pena = 0.0;
        const double vMaxSqr = pp.vMax * pp.vMax;
        const double thrAccMinSqr = pp.thrAccMin * pp.thrAccMin;
        const double thrAccMaxSqr = pp.thrAccMax * pp.thrAccMax;
        const double bdrMaxSqr = pp.bdrMax * pp.bdrMax;
        const V3 ell{{pp.ellipsoid[0], pp.ellipsoid[1], pp.ellipsoid[2]}};

        for (int i = 0; i < N; i++) {
            const double *c = &b[(size_t)i * 18];            // c(k, d) = c[k*3+d]
            double *gC = &gdCacc[(size_t)i * 18];
            const double step = T1[i] / cons[i];
            double s1 = 0.0;
            const int innerLoop = cons[i] + 1;
            for (int j = 0; j < innerLoop; j++) {
                if (!s1_accumulate) s1 = step * j;
                const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
                const double beta0[6] = {1.0, s1, s2, s3, s4, s5};
                const double beta1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
                const double beta2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
                const double beta3[6] = {0.0, 0.0, 0.0, 6.0, 24.0 * s1, 60.0 * s2};
                const double beta4[6] = {0.0, 0.0, 0.0, 0.0, 24.0, 120.0 * s1};
                const double alpha = 1.0 / cons[i] * j;
                V3 pos{{0, 0, 0}}, vel{{0, 0, 0}}, acc{{0, 0, 0}}, jer{{0, 0, 0}}, sna{{0, 0, 0}};
                for (int d = 0; d < 3; d++) {
                    double p = 0, v = 0, a = 0, jj = 0, sn = 0;
                    for (int k = 0; k < 6; k++) {
                        p += c[k * 3 + d] * beta0[k];
                        v += c[k * 3 + d] * beta1[k];
                        a += c[k * 3 + d] * beta2[k];
                        jj += c[k * 3 + d] * beta3[k];
                        sn += c[k * 3 + d] * beta4[k];
                    }
                    pos(d) = p; vel(d) = v; acc(d) = a; jer(d) = jj; sna(d) = sn;
                }

                V3 h = acc;
                h(2) += pp.gAcc;
                V3 zB, yB; M3 dzB, dnczB;
                normalizeFDF(h, zB, dzB);
                V3 czB{{0.0, zB(2), -zB(1)}};
                M3 cdzB;
                for (int q = 0; q < 3; q++) { cdzB.m[0][q] = 0.0; cdzB.m[1][q] = dzB.m[2][q]; cdzB.m[2][q] = -dzB.m[1][q]; }
                normalizeFDF(czB, yB, dnczB);
                V3 xB = cross3(yB, zB);
                M3 dyB = matmul(dnczB, cdzB);
                M3 dxB;
                for (int q = 0; q < 3; q++) {
                    V3 a1 = cross3(col(dyB, q), zB), a2 = cross3(yB, col(dzB, q));
                    for (int r = 0; r < 3; r++) dxB.m[r][q] = a1(r) + a2(r);
                }
                M3 rotM;
                for (int r = 0; r < 3; r++) { rotM.m[r][0] = xB(r); rotM.m[r][1] = yB(r); rotM.m[r][2] = zB(r); }

                V3 gTx = matvec(dxB, jer), gTy = matvec(dyB, jer), gTz = matvec(dzB, jer);   // gradSdTxyz cols

                const double fThr = std::sqrt(dot3(h, h));
                V3 dfThr{{h(0) / fThr, h(1) / fThr, h(2) / fThr}};
                const double sqrMagThr = fThr * fThr;
                V3 dSqrMagThr{{2 * h(0), 2 * h(1), 2 * h(2)}};
                V3 rotTrDotJer = matTvec(rotM, jer);
                V3 bdr{{rotTrDotJer(0) / fThr, rotTrDotJer(1) / fThr, rotTrDotJer(2) / fThr}};
                V3 xyBdr{{-bdr(1), bdr(0), 0.0}};
                const double sqrMagBdr = dot3(xyBdr, xyBdr);
                // dBdr = -rotTrDotJer dfThr^T / fThr^2 - rotM^T (dxB r0 + dyB r1 + dzB r2) / fThr   (CPU.hpp:293-294)
                M3 comb, dBdr;
                for (int r = 0; r < 3; r++)
                    for (int q = 0; q < 3; q++)
                        comb.m[r][q] = dxB.m[r][q] * rotTrDotJer(0) + dyB.m[r][q] * rotTrDotJer(1) + dzB.m[r][q] * rotTrDotJer(2);
                for (int r = 0; r < 3; r++)
                    for (int q = 0; q < 3; q++) {
                        double rt = rotM.m[0][r] * comb.m[0][q] + rotM.m[1][r] * comb.m[1][q] + rotM.m[2][r] * comb.m[2][q];
                        dBdr.m[r][q] = -rotTrDotJer(r) * dfThr(q) / (fThr * fThr) - rt / fThr;
                    }
                V3 dSqrMagBdr, dJerSqrMagBdr;
                // dxyBdr rows: (-dBdr.row(1), dBdr.row(0), 0); dSqrMagBdr = 2 xyBdr^T dxyBdr (CPU.hpp:295-296)
                for (int q = 0; q < 3; q++)
                    dSqrMagBdr(q) = 2.0 * xyBdr(0) * (-dBdr.m[1][q]) + 2.0 * xyBdr(1) * dBdr.m[0][q] + 2.0 * xyBdr(2) * 0.0;
                // dJerBdr = rotM^T / fThr (CPU.hpp:297-299)
                for (int q = 0; q < 3; q++) {
                    double dJ0 = rotM.m[q][0] / fThr, dJ1 = rotM.m[q][1] / fThr;   // dJerBdr(0,q), dJerBdr(1,q)
                    dJerSqrMagBdr(q) = 2.0 * xyBdr(0) * (-dJ1) + 2.0 * xyBdr(1) * dJ0 + 2.0 * xyBdr(2) * 0.0;
                }

                const double violaVel = dot3(vel, vel) - vMaxSqr;
                const double violaThrl = thrAccMinSqr - sqrMagThr;
                const double violaThrh = sqrMagThr - thrAccMaxSqr;
                const double violaBdr = sqrMagBdr - bdrMaxSqr;

                const double omg = (j == 0 || j == innerLoop - 1) ? 0.5 : 1.0;

                const int idx = idxHs[i];
                const std::vector<double> &H = cfgHs[idx];
                const int K = (int)(H.size() / 6);
                for (int k = 0; k < K; k++) {
                    V3 n{{H[6 * k], H[6 * k + 1], H[6 * k + 2]}};
                    V3 point{{H[6 * k + 3], H[6 * k + 4], H[6 * k + 5]}};
                    V3 gradSdT{{dot3(gTx, n), dot3(gTy, n), dot3(gTz, n)}};
                    const double outerNormaldVel = dot3(n, vel);
                    // gradSdC{x,y,z} = (beta2 n^T) d{x,y,z}B  →  beta2 (n^T dB)   (CPU.hpp:317-320, unconditional)
                    V3 nx = matTvec(dxB, n), ny = matTvec(dyB, n), nz = matTvec(dzB, n);
                    double gradSdCx[6][3], gradSdCy[6][3], gradSdCz[6][3];
                    for (int r = 0; r < 6; r++)
                        for (int d = 0; d < 3; d++) {
                            gradSdCx[r][d] = beta2[r] * nx(d);
                            gradSdCy[r][d] = beta2[r] * ny(d);
                            gradSdCz[r][d] = beta2[r] * nz(d);
                        }
                    V3 eNormGd = matTvec(rotM, n);
                    for (int d = 0; d < 3; d++) eNormGd(d) *= ell(d);
                    const double eNorm = std::sqrt(dot3(eNormGd, eNormGd));
                    for (int d = 0; d < 3; d++) eNormGd(d) /= eNorm;
                    V3 dp{{pos(0) - point(0), pos(1) - point(1), pos(2) - point(2)}};
                    double signedDist = dot3(n, dp) + eNorm;
                    for (int d = 0; d < 3; d++) eNormGd(d) *= ell(d);
                    signedDist += pp.safeMargin;
                    if (signedDist > 0) {
                        const double signedDistSqr = signedDist * signedDist;
                        const double signedDistCub = signedDist * signedDistSqr;
                        const double gradSignedDt = alpha * (outerNormaldVel + gradSdT(0) * eNormGd(0) +
                                                             gradSdT(1) * eNormGd(1) + gradSdT(2) * eNormGd(2));
                        const double w = omg * step * pp.ci[0] * 3.0 * signedDistSqr;
                        for (int r = 0; r < 6; r++)
                            for (int d = 0; d < 3; d++) {
                                const double gradSdC = beta0[r] * n(d) + gradSdCx[r][d] * eNormGd(0) +
                                                       gradSdCy[r][d] * eNormGd(1) + gradSdCz[r][d] * eNormGd(2);
                                gC[r * 3 + d] += w * gradSdC;
                            }
                        gdT[i] += omg * pp.ci[0] * (3.0 * signedDistSqr * gradSignedDt * step + signedDistCub / cons[i]);
                        pena += omg * step * pp.ci[0] * signedDistCub;
                    }
                }

                if (violaVel > 0.0) {                                   // CPU.hpp:347-359
                    double violaVelPenaD = violaVel * violaVel;
                    const double violaVelPena = violaVelPenaD * violaVel;
                    violaVelPenaD *= 3.0;
                    const double gradViolaVt = 2.0 * alpha * dot3(vel, acc);
                    const double w = omg * step * pp.ci[1] * violaVelPenaD;
                    for (int r = 0; r < 6; r++)
                        for (int d = 0; d < 3; d++) gC[r * 3 + d] += w * (2.0 * beta1[r] * vel(d));
                    gdT[i] += omg * (pp.ci[1] * violaVelPenaD * gradViolaVt * step + pp.ci[1] * violaVelPena / cons[i]);
                    pena += omg * step * pp.ci[1] * violaVelPena;
                }
                if (violaThrl > 0.0) {                                  // CPU.hpp:361-372
                    double violaThrlPenaD = violaThrl * violaThrl;
                    const double violaThrlPena = violaThrlPenaD * violaThrl;
                    violaThrlPenaD *= 3.0;
                    const double gradViolaThrlt = -alpha * dot3(dSqrMagThr, jer);
                    const double w = omg * step * pp.ci[2] * violaThrlPenaD;
                    for (int r = 0; r < 6; r++)
                        for (int d = 0; d < 3; d++) gC[r * 3 + d] += w * (-beta2[r] * dSqrMagThr(d));
                    gdT[i] += omg * (pp.ci[2] * violaThrlPenaD * gradViolaThrlt * step + pp.ci[2] * violaThrlPena / cons[i]);
                    pena += omg * step * pp.ci[2] * violaThrlPena;
                }
                if (violaThrh > 0.0) {                                  // CPU.hpp:374-385 (weight index 2 again)
                    double violaThrhPenaD = violaThrh * violaThrh;
                    const double violaThrhPena = violaThrhPenaD * violaThrh;
                    violaThrhPenaD *= 3.0;
                    const double gradViolaThrht = alpha * dot3(dSqrMagThr, jer);
                    const double w = omg * step * pp.ci[2] * violaThrhPenaD;
                    for (int r = 0; r < 6; r++)
                        for (int d = 0; d < 3; d++) gC[r * 3 + d] += w * (beta2[r] * dSqrMagThr(d));
                    gdT[i] += omg * (pp.ci[2] * violaThrhPenaD * gradViolaThrht * step + pp.ci[2] * violaThrhPena / cons[i]);
                    pena += omg * step * pp.ci[2] * violaThrhPena;
                }
                if (violaBdr > 0.0) {                                   // CPU.hpp:387-398
                    double violaBdrPenaD = violaBdr * violaBdr;
                    const double violaBdrPena = violaBdrPenaD * violaBdr;
                    violaBdrPenaD *= 3.0;
                    const double gradViolaBdrt = alpha * (dot3(dSqrMagBdr, jer) + dot3(dJerSqrMagBdr, sna));
                    const double w = omg * step * pp.ci[3] * violaBdrPenaD;
                    for (int r = 0; r < 6; r++)
                        for (int d = 0; d < 3; d++)
                            gC[r * 3 + d] += w * (beta2[r] * dSqrMagBdr(d) + beta3[r] * dJerSqrMagBdr(d));
                    gdT[i] += omg * (pp.ci[3] * violaBdrPenaD * gradViolaBdrt * step + pp.ci[3] * violaBdrPena / cons[i]);
                    pena += omg * step * pp.ci[3] * violaBdrPena;
                }
                s1 += step;
            }
        }
        cost += pena;
    }

    // CPU.hpp:522-553
    void evalTrajCostGrad(const int *cons, const int *idxHs, const std::vector<std::vector<double>> &cfgHs,
                          const PenaltyParams &pp, double &cost, double *gdT, double *gdInPs) {
        std::fill(gdT, gdT + N, 0.0);
        std::fill(gdInPs, gdInPs + 3 * (N - 1), 0.0);
        std::fill(gdC.begin(), gdC.end(), 0.0);
        cost = getTrajJerkCost();
        addGradJbyT(gdT);
        addGradJbyC(gdC.data());
        addTimeIntPenalty(cons, idxHs, cfgHs, pp, cost, gdT, gdC.data());
        A.solveAdj(gdC.data(), 3);
        addPropCtoT(gdC.data(), gdT);
        addPropCtoP(gdC.data(), gdInPs);
    }
};

// ------------------------------------------------------------------------------------------
// SE3GCOPTER — CPU.hpp:567-1269
// ------------------------------------------------------------------------------------------
struct Config {                  // mirrors SE3GCOPTER::setup's scalar arguments, CPU.hpp:1076-1092
    double rho, total_t, grid_res;
    int qd_intervals, c2_diffeo;
    double horiz_half_len, vert_half_len, safe_margin;
    double vel_max, thr_acc_min, thr_acc_max, body_rate_max, grav_acc;
    double penalty_pvtb[4];
};

struct Problem {
    bool c2dfm, softT;
    double rho, sumT;
    MincoS3 jerkOpt;
    double iState[9], fState[9];             // column-major 3x3 (p | v | a)
    std::vector<std::vector<double>> cfgVs;  // 3 x nv col-major: [v0, v1-v0, ...] (CPU.hpp:1049)
    std::vector<std::vector<double>> cfgHs;  // 6 x K col-major
    std::vector<double> gdInPs;
    std::vector<int> intervals, idxVs, idxHs, cons;
    int coarseN, fineN, dimFreeT, dimFreeP;
    std::vector<double> coarseT, fineT, innerP;
    PenaltyParams pp;
    long evals = 0;
    std::vector<double> trace;     // every point the optimiser evaluated (n doubles each), when trace_cap > 0
    long trace_cap = 0;

    static inline int nvOf(const std::vector<double> &V) { return (int)(V.size() / 3); }

    // CPU.hpp:626-676
    static void forwardT(const double *t, int tn, std::vector<double> &vecT, bool soft, double sT, bool c2) {
        if (soft) {
            int M = (int)vecT.size();
            for (int i = 0; i < M; i++)
                vecT[i] = c2 ? (t[i] > 0.0 ? ((0.5 * t[i] + 1.0) * t[i] + 1.0) : 1.0 / ((0.5 * t[i] - 1.0) * t[i] + 1.0))
                             : std::exp(t[i]);
        } else {
            int Ms1 = tn;
            for (int i = 0; i < Ms1; i++)
                vecT[i] = c2 ? (t[i] > 0.0 ? ((0.5 * t[i] + 1.0) * t[i] + 1.0) : 1.0 / ((0.5 * t[i] - 1.0) * t[i] + 1.0))
                             : std::exp(t[i]);
            vecT[Ms1] = 0.0;
            double sum = 0.0;
            for (int i = 0; i <= Ms1; i++) sum += vecT[i];
            const double den = 1.0 + sum;
            for (int i = 0; i <= Ms1; i++) vecT[i] /= den;
            sum = 0.0;
            for (int i = 0; i <= Ms1; i++) sum += vecT[i];
            vecT[Ms1] = 1.0 - sum;
            for (int i = 0; i <= Ms1; i++) vecT[i] *= sT;
        }
    }
    // CPU.hpp:679-726
    static void backwardT(const std::vector<double> &vecT, double *t, int tn, bool soft, bool c2) {
        if (soft) {
            int M = (int)vecT.size();
            for (int i = 0; i < M; i++)
                t[i] = c2 ? (vecT[i] > 1.0 ? (std::sqrt(2.0 * vecT[i] - 1.0) - 1.0) : (1.0 - std::sqrt(2.0 / vecT[i] - 1.0)))
                          : std::log(vecT[i]);
        } else {
            int Ms1 = tn;
            for (int i = 0; i < Ms1; i++) {
                double r = vecT[i] / vecT[Ms1];
                t[i] = c2 ? (r > 1.0 ? (std::sqrt(2.0 * r - 1.0) - 1.0) : (1.0 - std::sqrt(2.0 / r - 1.0))) : std::log(r);
            }
        }
    }
    // CPU.hpp:729-747
    static void forwardP(const double *p, const std::vector<int> &idVs, const std::vector<std::vector<double>> &cfgPolyVs,
                         std::vector<double> &inP) {
        int M = (int)(inP.size() / 3);
        int j = 0;
        std::vector<double> q;
        for (int i = 0; i < M; i++) {
            const std::vector<double> &V = cfgPolyVs[idVs[i]];
            int k = nvOf(V) - 1;
            double nrm = 0.0;
            for (int a = 0; a < k; a++) nrm += p[j + a] * p[j + a];
            const double sc = 2.0 / (1.0 + nrm);
            q.resize(k);
            for (int a = 0; a < k; a++) q[a] = sc * p[j + a];
            for (int r = 0; r < 3; r++) {
                double s = 0.0;
                for (int a = 0; a < k; a++) s += V[3 * (a + 1) + r] * (q[a] * q[a]);
                inP[i * 3 + r] = s + V[r];
            }
            j += k;
        }
    }
    // CPU.hpp:749-774 — pobs = [target, v0, edges...] (3 x (n+2) col-major)
    static double objectiveNLS(void *ptrPOBs, const double *x, double *grad, const int n) {
        const double *pobs = (const double *)ptrPOBs;
        double qnsqr = 0.0;
        for (int a = 0; a < n; a++) qnsqr += x[a] * x[a];
        const double qnsqrp1 = qnsqr + 1.0;
        const double qnsqrp1sqr = qnsqrp1 * qnsqrp1;
        std::vector<double> r(n), gdr(n);
        const double sc = 2.0 / qnsqrp1;
        for (int a = 0; a < n; a++) r[a] = sc * x[a];
        double delta[3];
        for (int q = 0; q < 3; q++) {
            double s = 0.0;
            for (int a = 0; a < n; a++) s += pobs[3 * (a + 2) + q] * (r[a] * r[a]);
            delta[q] = s + pobs[3 + q] - pobs[q];
        }
        const double cost = delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2];
        const double gradR3[3] = {2 * delta[0], 2 * delta[1], 2 * delta[2]};
        for (int a = 0; a < n; a++) {
            double s = pobs[3 * (a + 2)] * gradR3[0] + pobs[3 * (a + 2) + 1] * gradR3[1] + pobs[3 * (a + 2) + 2] * gradR3[2];
            gdr[a] = s * r[a] * 2.0;
        }
        double gdrDotP = 0.0;
        for (int a = 0; a < n; a++) gdrDotP += gdr[a] * x[a];
        for (int a = 0; a < n; a++) grad[a] = gdr[a] * 2.0 / qnsqrp1 - x[a] * 4.0 * gdrDotP / qnsqrp1sqr;
        return cost;
    }
    // CPU.hpp:777-813
    typedef int (*nls_solver)(int n, double *x, double *fx, orc_lbfgs::eval_fn fn, void *inst, const orc_lbfgs::Params &pm);
    static void backwardP(const std::vector<double> &inP, const std::vector<int> &idVs,
                          const std::vector<std::vector<double>> &cfgPolyVs, double *p, nls_solver solver) {
        int M = (int)(inP.size() / 3);
        int j = 0;
        orc_lbfgs::Params nls;
        nls.g_epsilon = FLT_EPSILON;
        nls.max_iterations = 128;
        std::vector<double> pobs;
        for (int i = 0; i < M; i++) {
            const std::vector<double> &V = cfgPolyVs[idVs[i]];
            int k = nvOf(V) - 1;
            for (int a = 0; a < k; a++) p[j + a] = 1.0 / (std::sqrt(k + 1.0) + 1.0);
            pobs.resize(3 * (k + 2));
            for (int r = 0; r < 3; r++) pobs[r] = inP[i * 3 + r];
            std::memcpy(&pobs[3], V.data(), sizeof(double) * 3 * (k + 1));
            double minSqrD;
            solver(k, p + j, &minSqrD, &Problem::objectiveNLS, pobs.data(), nls);
            j += k;
        }
    }
    // CPU.hpp:816-894
    static void addLayerTGrad(const double *t, int tn, std::vector<double> &gradT, bool soft, double sT, bool c2) {
        if (soft) {
            int M = tn;
            for (int i = 0; i < M; i++) {
                if (c2) {
                    if (t[i] > 0) gradT[i] *= t[i] + 1.0;
                    else {
                        double denSqrt = (0.5 * t[i] - 1.0) * t[i] + 1.0;
                        gradT[i] *= (1.0 - t[i]) / (denSqrt * denSqrt);
                    }
                } else gradT[i] *= std::exp(t[i]);
            }
        } else {
            int Ms1 = tn;
            std::vector<double> gFree(Ms1), dExpTau(Ms1);
            for (int i = 0; i < Ms1; i++) gFree[i] = sT * gradT[i];
            const double gTail = sT * gradT[Ms1];
            double expTauSum = 0.0, gFreeDotExpTau = 0.0;
            if (c2) {
                for (int i = 0; i < Ms1; i++) {
                    double expTau;
                    if (t[i] > 0) { expTau = (0.5 * t[i] + 1.0) * t[i] + 1.0; dExpTau[i] = t[i] + 1.0; }
                    else {
                        double denSqrt = (0.5 * t[i] - 1.0) * t[i] + 1.0;
                        expTau = 1.0 / denSqrt;
                        dExpTau[i] = (1.0 - t[i]) / (denSqrt * denSqrt);
                    }
                    expTauSum += expTau;
                    gFreeDotExpTau += expTau * gFree[i];
                }
            } else {
                for (int i = 0; i < Ms1; i++) { dExpTau[i] = std::exp(t[i]); expTauSum += dExpTau[i]; }
                for (int i = 0; i < Ms1; i++) gFreeDotExpTau += gFree[i] * dExpTau[i];
            }
            const double denSqrt = expTauSum + 1.0;
            for (int i = 0; i < Ms1; i++)
                gradT[i] = (gFree[i] - gTail) * dExpTau[i] / denSqrt -
                           (gFreeDotExpTau - gTail * expTauSum) * dExpTau[i] / (denSqrt * denSqrt);
            gradT[Ms1] = 0.0;
        }
    }
    // CPU.hpp:897-928
    static void addLayerPGrad(const double *p, const std::vector<int> &idVs, const std::vector<std::vector<double>> &cfgPolyVs,
                              const std::vector<double> &gradInPs, double *grad) {
        int M = (int)(gradInPs.size() / 3);
        int j = 0;
        std::vector<double> r, gdr;
        for (int i = 0; i < M; i++) {
            const std::vector<double> &V = cfgPolyVs[idVs[i]];
            int k = nvOf(V) - 1;
            double qnsqr = 0.0;
            for (int a = 0; a < k; a++) qnsqr += p[j + a] * p[j + a];
            const double qnsqrp1 = qnsqr + 1.0, qnsqrp1sqr = qnsqrp1 * qnsqrp1;
            r.resize(k); gdr.resize(k);
            const double sc = 2.0 / qnsqrp1;
            for (int a = 0; a < k; a++) r[a] = sc * p[j + a];
            for (int a = 0; a < k; a++) {
                double s = V[3 * (a + 1)] * gradInPs[i * 3] + V[3 * (a + 1) + 1] * gradInPs[i * 3 + 1] + V[3 * (a + 1) + 2] * gradInPs[i * 3 + 2];
                gdr[a] = s * r[a] * 2.0;
            }
            double gdrDotQ = 0.0;
            for (int a = 0; a < k; a++) gdrDotQ += gdr[a] * p[j + a];
            for (int a = 0; a < k; a++) grad[j + a] = gdr[a] * 2.0 / qnsqrp1 - p[j + a] * 4.0 * gdrDotQ / qnsqrp1sqr;
            j += k;
        }
    }
    // CPU.hpp:930-944
    static void splitToFineT(const std::vector<double> &cT, const std::vector<int> &intervs, std::vector<double> &fT) {
        int offset = 0;
        for (size_t i = 0; i < intervs.size(); i++) {
            int iv = intervs[i];
            for (int a = 0; a < iv; a++) fT[offset + a] = cT[i] / iv;
            offset += iv;
        }
    }
    // CPU.hpp:946-959
    static void mergeToCoarseGradT(const std::vector<int> &intervs, std::vector<double> &fineGdT) {
        int offset = 0;
        for (size_t i = 0; i < intervs.size(); i++) {
            int iv = intervs[i];
            double s = 0.0;
            for (int a = 0; a < iv; a++) s += fineGdT[offset + a];
            fineGdT[i] = s / iv;
            offset += iv;
        }
    }

    // CPU.hpp:961-1000
    static double objectiveFunc(void *ptrObj, const double *x, double *grad, const int n) {
        Problem &obj = *(Problem *)ptrObj;
        if ((long)(obj.trace.size() / (size_t)n) < obj.trace_cap) obj.trace.insert(obj.trace.end(), x, x + n);
        obj.evals++;
        const int dimT = obj.dimFreeT;
        const double *t = x, *p = x + dimT;
        std::vector<double> proxyGradT(obj.fineN);
        forwardT(t, dimT, obj.coarseT, obj.softT, obj.sumT, obj.c2dfm);
        splitToFineT(obj.coarseT, obj.intervals, obj.fineT);
        forwardP(p, obj.idxVs, obj.cfgVs, obj.innerP);
        double cost;
        obj.jerkOpt.generate(obj.innerP.data(), obj.fineT.data());
        obj.jerkOpt.evalTrajCostGrad(obj.cons.data(), obj.idxHs.data(), obj.cfgHs, obj.pp, cost, proxyGradT.data(), obj.gdInPs.data());
        double sumT = 0.0;
        for (int i = 0; i < obj.coarseN; i++) sumT += obj.coarseT[i];
        cost += obj.rho * sumT;
        for (int i = 0; i < obj.fineN; i++) proxyGradT[i] += obj.rho;
        mergeToCoarseGradT(obj.intervals, proxyGradT);
        addLayerTGrad(t, dimT, proxyGradT, obj.softT, obj.sumT, obj.c2dfm);
        addLayerPGrad(p, obj.idxVs, obj.cfgVs, obj.gdInPs, grad + dimT);
        for (int i = 0; i < dimT; i++) grad[i] = proxyGradT[i];
        return cost;
    }

    // vertex mean of V-polytope m: rightCols(k).rowwise().sum() / (1+k) + col(0)   (CPU.hpp:1018-1019, 1206-1207)
    void polyCentre(int m, double *c) const {
        const std::vector<double> &V = cfgVs[m];
        int k = nvOf(V) - 1;
        for (int r = 0; r < 3; r++) {
            double s = 0.0;
            for (int a = 0; a < k; a++) s += V[3 * (a + 1) + r];
            c[r] = s / (1.0 + k) + V[r];
        }
    }
    // CPU.hpp:1003-1029
    void gridMesh(double gridResolution, std::vector<int> &iv) const {
        int M = (int)iv.size();
        double lastP[3], curP[3] = {iState[0], iState[1], iState[2]};
        for (int i = 0; i < M - 1; i++) {
            std::memcpy(lastP, curP, sizeof(curP));
            polyCentre(2 * i + 1, curP);
            double dn = std::sqrt((curP[0] - lastP[0]) * (curP[0] - lastP[0]) + (curP[1] - lastP[1]) * (curP[1] - lastP[1]) +
                                  (curP[2] - lastP[2]) * (curP[2] - lastP[2]));
            int cur = (int)std::ceil(dn / gridResolution);
            iv[i] = cur > 0 ? cur : 1;
        }
        std::memcpy(lastP, curP, sizeof(curP));
        curP[0] = fState[0]; curP[1] = fState[1]; curP[2] = fState[2];
        double dn = std::sqrt((curP[0] - lastP[0]) * (curP[0] - lastP[0]) + (curP[1] - lastP[1]) * (curP[1] - lastP[1]) +
                              (curP[2] - lastP[2]) * (curP[2] - lastP[2]));
        int cur = (int)std::ceil(dn / gridResolution);
        iv[M - 1] = cur > 0 ? cur : 1;
    }

    // CPU.hpp:1076-1186.  extractVs (CPU.hpp:1031-1074) is replaced by its OUTPUT: the caller
    // supplies the vertices of the 2*coarseN-1 polytopes (SURVEY.md §8f-f1 "next" row); the
    // [v0, v_r - v0] re-basing of CPU.hpp:1049 is done here.
    bool setup(const Config &cf, const double *iniState, const double *finState, int coarseN_, const int *hOff,
               const double *hRec, const int *vOff, const double *vRec) {
        c2dfm = cf.c2_diffeo != 0;
        softT = cf.rho > 0;
        if (softT) { rho = cf.rho; sumT = 1.0; } else { rho = 0.0; sumT = cf.total_t; }
        std::memcpy(iState, iniState, sizeof(iState));
        std::memcpy(fState, finState, sizeof(fState));
        coarseN = coarseN_;
        cfgHs.assign(coarseN, {});
        for (int i = 0; i < coarseN; i++) {
            cfgHs[i].assign(hRec + 6 * (size_t)hOff[i], hRec + 6 * (size_t)hOff[i + 1]);
            int K = hOff[i + 1] - hOff[i];
            for (int k = 0; k < K; k++) {                          // topRows<3>().colwise().normalize(), CPU.hpp:1116
                double *n = &cfgHs[i][6 * k];
                double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                n[0] /= nn; n[1] /= nn; n[2] /= nn;
            }
        }
        cfgVs.assign(2 * coarseN - 1, {});
        for (int m = 0; m < 2 * coarseN - 1; m++) {
            int nv = vOff[m + 1] - vOff[m];
            if (nv < 1) return false;                               // empty interior → setup fails (CPU.hpp:1118-1121)
            cfgVs[m].resize(3 * (size_t)nv);
            const double *v = vRec + 3 * (size_t)vOff[m];
            for (int r = 0; r < 3; r++) cfgVs[m][r] = v[r];
            for (int a = 1; a < nv; a++)
                for (int r = 0; r < 3; r++) cfgVs[m][3 * a + r] = v[3 * a + r] - v[r];
        }
        intervals.assign(coarseN, 0);
        gridMesh(cf.grid_res, intervals);
        fineN = 0;
        for (int i = 0; i < coarseN; i++) fineN += intervals[i];
        cons.assign(fineN, cf.qd_intervals);
        idxVs.assign(fineN - 1, 0);
        idxHs.assign(fineN, 0);
        dimFreeT = softT ? coarseN : coarseN - 1;
        dimFreeP = 0;
        int offset = 0;
        for (int i = 0; i < coarseN; i++) {
            int interval = intervals[i];
            for (int j = 0; j < interval; j++) {
                if (j < interval - 1) { idxVs[offset] = 2 * i; dimFreeP += nvOf(cfgVs[2 * i]) - 1; }
                else if (i < coarseN - 1) { idxVs[offset] = 2 * i + 1; dimFreeP += nvOf(cfgVs[2 * i + 1]) - 1; }
                idxHs[offset] = i;
                offset++;
            }
        }
        for (int q = 0; q < 4; q++) pp.ci[q] = cf.penalty_pvtb[q];
        pp.ellipsoid[0] = cf.horiz_half_len; pp.ellipsoid[1] = cf.horiz_half_len; pp.ellipsoid[2] = cf.vert_half_len;
        pp.safeMargin = cf.safe_margin; pp.vMax = cf.vel_max; pp.thrAccMin = cf.thr_acc_min; pp.thrAccMax = cf.thr_acc_max;
        pp.bdrMax = cf.body_rate_max; pp.gAcc = cf.grav_acc;
        // legal initial speed on the COPIES (CPU.hpp:1166-1170); jerkOpt.reset gets the unclipped states (CPU.hpp:1180)
        double tn = std::sqrt(iState[3] * iState[3] + iState[4] * iState[4] + iState[5] * iState[5]);
        double sc = tn > pp.vMax ? (pp.vMax / tn) : 1.0;
        for (int r = 0; r < 3; r++) iState[3 + r] *= sc;
        tn = std::sqrt(fState[3] * fState[3] + fState[4] * fState[4] + fState[5] * fState[5]);
        sc = tn > pp.vMax ? (pp.vMax / tn) : 1.0;
        for (int r = 0; r < 3; r++) fState[3 + r] *= sc;
        coarseT.assign(coarseN, 0.0);
        fineT.assign(fineN, 0.0);
        innerP.assign(3 * (size_t)(fineN - 1), 0.0);
        gdInPs.assign(3 * (size_t)(fineN - 1), 0.0);
        jerkOpt.reset(iniState, finState, fineN);
        return true;
    }

    // CPU.hpp:1188-1228
    void setInitial(std::vector<double> &vecT, std::vector<double> &vecInP) const {
        const double maxSpeedForAllocation = 10.0;
        int M = (int)vecT.size();
        double lastP[3], curP[3] = {iState[0], iState[1], iState[2]}, delta[3];
        int offset = 0;
        for (int i = 0; i < M - 1; i++) {
            std::memcpy(lastP, curP, sizeof(curP));
            int interv = intervals[i];
            polyCentre(2 * i + 1, curP);
            for (int r = 0; r < 3; r++) delta[r] = curP[r] - lastP[r];
            vecT[i] = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]) / std::min(pp.vMax, maxSpeedForAllocation);
            for (int r = 0; r < 3; r++) delta[r] /= interv;
            for (int j = 0; j < interv; j++) {
                for (int r = 0; r < 3; r++) vecInP[offset * 3 + r] = (j + 1) * delta[r] + lastP[r];
                offset++;
            }
        }
        int interv = intervals[M - 1];
        std::memcpy(lastP, curP, sizeof(curP));
        curP[0] = fState[0]; curP[1] = fState[1]; curP[2] = fState[2];
        for (int r = 0; r < 3; r++) delta[r] = curP[r] - lastP[r];
        vecT[M - 1] = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]) / std::min(pp.vMax, maxSpeedForAllocation);
        for (int r = 0; r < 3; r++) delta[r] /= interv;
        for (int j = 0; j < interv - 1; j++) {
            for (int r = 0; r < 3; r++) vecInP[offset * 3 + r] = (j + 1) * delta[r] + lastP[r];
            offset++;
        }
    }

    // first half of optimize(), CPU.hpp:1233-1240
    void initialGuess(double *x, nls_solver solver) {
        setInitial(coarseT, innerP);
        backwardT(coarseT, x, dimFreeT, softT, c2dfm);
        backwardP(innerP, idxVs, cfgVs, x + dimFreeT, solver);
    }
    // x -> (T fine, innerP) -> generate; CPU.hpp:1258-1262
    void finalGenerate(const double *x) {
        forwardT(x, dimFreeT, coarseT, softT, sumT, c2dfm);
        splitToFineT(coarseT, intervals, fineT);
        forwardP(x + dimFreeT, idxVs, cfgVs, innerP);
        jerkOpt.generate(innerP.data(), fineT.data());
    }
};

static int port_solver(int n, double *x, double *fx, orc_lbfgs::eval_fn fn, void *inst, const orc_lbfgs::Params &pm) {
    return orc_lbfgs::optimize(n, x, fx, fn, inst, pm);
}

} // namespace orc

// ------------------------------------------------------------------------------------------
// C entry points (ctypes).  An external L-BFGS (oracle/_ref: the reference's own lbfgs.hpp)
// may be plugged in through orc_set_lbfgs().
// ------------------------------------------------------------------------------------------
extern "C" {

// signature of the optimiser hook: params = {mem_size, g_epsilon, past, delta, max_iterations,
// max_linesearch, min_step, max_step, f_dec_coeff, s_curv_coeff, xtol} as doubles
typedef int (*orc_lbfgs_hook)(int n, double *x, double *fx, orc_lbfgs::eval_fn fn, void *inst, const double *params11);
static orc_lbfgs_hook g_hook = nullptr;

static int hook_solver(int n, double *x, double *fx, orc_lbfgs::eval_fn fn, void *inst, const orc_lbfgs::Params &pm) {
    const double p[11] = {(double)pm.mem_size, pm.g_epsilon, (double)pm.past, pm.delta, (double)pm.max_iterations,
                          (double)pm.max_linesearch, pm.min_step, pm.max_step, pm.f_dec_coeff, pm.s_curv_coeff, pm.xtol};
    return g_hook(n, x, fx, fn, inst, p);
}
static orc::Problem::nls_solver current_solver() { return g_hook ? &hook_solver : &orc::port_solver; }

void orc_set_lbfgs(orc_lbfgs_hook h) { g_hook = h; }

void *orc_create(const orc::Config *cfg, const double *iniState, const double *finState, int coarseN, const int *hOff,
                 const double *hRec, const int *vOff, const double *vRec) {
    orc::Problem *p = new orc::Problem();
    if (!p->setup(*cfg, iniState, finState, coarseN, hOff, hRec, vOff, vRec)) { delete p; return nullptr; }
    return p;
}
void orc_destroy(void *h) { delete (orc::Problem *)h; }
void orc_dims(void *h, int *out4) {
    orc::Problem *p = (orc::Problem *)h;
    out4[0] = p->coarseN; out4[1] = p->fineN; out4[2] = p->dimFreeT; out4[3] = p->dimFreeP;
}
void orc_set_abscissa_mode(void *h, int accumulate) { ((orc::Problem *)h)->jerkOpt.s1_accumulate = accumulate != 0; }
void orc_get_maps(void *h, int *intervals, int *idxVs, int *idxHs) {
    orc::Problem *p = (orc::Problem *)h;
    std::copy(p->intervals.begin(), p->intervals.end(), intervals);
    std::copy(p->idxVs.begin(), p->idxVs.end(), idxVs);
    std::copy(p->idxHs.begin(), p->idxHs.end(), idxHs);
}
void orc_initial_guess(void *h, double *x) { ((orc::Problem *)h)->initialGuess(x, current_solver()); }
// setInitial only: coarse T (coarseN) and inner waypoints (3 x (fineN-1) col-major)
void orc_set_initial(void *h, double *T, double *P) {
    orc::Problem *p = (orc::Problem *)h;
    p->setInitial(p->coarseT, p->innerP);
    std::copy(p->coarseT.begin(), p->coarseT.end(), T);
    std::copy(p->innerP.begin(), p->innerP.end(), P);
}
double orc_objective(void *h, const double *x, double *g) {
    orc::Problem *p = (orc::Problem *)h;
    return orc::Problem::objectiveFunc(p, x, g, p->dimFreeT + p->dimFreeP);
}
// x -> fine T (fineN), inner points (3 x (fineN-1)), coefficients (6 fineN x 3 row-major)
void orc_forward(void *h, const double *x, double *T, double *P, double *C) {
    orc::Problem *p = (orc::Problem *)h;
    p->finalGenerate(x);
    if (T) std::copy(p->fineT.begin(), p->fineT.end(), T);
    if (P) std::copy(p->innerP.begin(), p->innerP.end(), P);
    if (C) std::copy(p->jerkOpt.b.begin(), p->jerkOpt.b.end(), C);
}
// inverse maps of the diffeomorphisms (tests: forward∘backward = id)
void orc_backward(void *h, const double *Tcoarse, const double *P, double *x) {
    orc::Problem *p = (orc::Problem *)h;
    std::vector<double> vT(Tcoarse, Tcoarse + p->coarseN), vP(P, P + 3 * (p->fineN - 1));
    orc::Problem::backwardT(vT, x, p->dimFreeT, p->softT, p->c2dfm);
    orc::Problem::backwardP(vP, p->idxVs, p->cfgVs, x + p->dimFreeT, current_solver());
}
// MINCO generate only (a5): inPs 3 x (N-1) col-major, T[N] → C (6N x 3 row-major)
void orc_generate(void *h, const double *inPs, const double *T, double *C) {
    orc::Problem *p = (orc::Problem *)h;
    p->jerkOpt.generate(inPs, T);
    std::copy(p->jerkOpt.b.begin(), p->jerkOpt.b.end(), C);
}
// dense copy of the band AFTER generate (the LU factors) and a dense re-assembly of A before
// factorisation, for the dense-solve cross-checks
void orc_dense_A(void *h, const double *T, double *Adense /* (6N)^2 row-major */) {
    orc::Problem *p = (orc::Problem *)h;
    orc::MincoS3 m;
    m.reset(p->jerkOpt.headPVA, p->jerkOpt.tailPVA, p->fineN);
    // assemble exactly as generate() does, but stop before factorizeLU: replay into a dense matrix
    std::vector<double> zeros(3 * (size_t)(p->fineN - 1), 0.0);
    int n6 = 6 * p->fineN;
    // generate() factorises in place, so rebuild A by multiplying L*U back
    m.generate(zeros.data(), T);
    std::fill(Adense, Adense + (size_t)n6 * n6, 0.0);
    for (int i = 0; i < n6; i++)
        for (int j = std::max(0, i - 6); j <= std::min(n6 - 1, i + 6); j++) {
            double s = 0.0;
            for (int k = 0; k <= std::min(i, j); k++) {
                if (i - k > 6 || j - k > 6) continue;
                double l = (k == i) ? 1.0 : m.A.at(i, k);
                s += l * m.A.at(k, j);
            }
            Adense[(size_t)i * n6 + j] = s;
        }
}
// in-place A^T \ rhs with the factors left by the last generate()/objective (a8)
void orc_solve_adj(void *h, double *rhs /* 6N x 3 row-major */) { ((orc::Problem *)h)->jerkOpt.A.solveAdj(rhs, 3); }
void orc_solve(void *h, double *rhs) { ((orc::Problem *)h)->jerkOpt.A.solve(rhs, 3); }
// a8 + a9 alone: with the factors left by the last generate(), lambda = A^-T gdC, then addPropCtoT / addPropCtoP
// (CPU.hpp:97-161).  gdC is overwritten with lambda; gdT (N) and gdP (3 x (N-1) col-major) are ACCUMULATED into.
void orc_backprop(void *h, double *gdC, double *gdT, double *gdP) {
    orc::Problem *p = (orc::Problem *)h;
    p->jerkOpt.A.solveAdj(gdC, 3);
    p->jerkOpt.addPropCtoT(gdC, gdT);
    p->jerkOpt.addPropCtoP(gdC, gdP);
}
double orc_jerk_cost(void *h) { return ((orc::Problem *)h)->jerkOpt.getTrajJerkCost(); }

// a7 alone with cuda_computer::compute semantics (accumulates into cost/gdT/gdC — cuda_computer.cu:551-558)
void orc_penalty(void *h, const double *T, const double *C, double *cost, double *gdT, double *gdC) {
    orc::Problem *p = (orc::Problem *)h;
    orc::MincoS3 &m = p->jerkOpt;
    m.T1.assign(T, T + m.N);
    std::copy(C, C + (size_t)18 * m.N, m.b.begin());
    m.addTimeIntPenalty(p->cons.data(), p->idxHs.data(), p->cfgHs, p->pp, *cost, gdT, gdC);
}

// optimize(): CPU.hpp:1230-1268.  Returns the L-BFGS status (the reference discards it,
// CPU.hpp:1249); *jerk_cost is what the reference returns (CPU.hpp:1267).
int orc_optimize(void *h, double relCostTol, int max_iterations, double *x_inout, int use_x_as_start, double *C, double *T,
                 double *jerk_cost, double *final_obj, long *n_evals, int *n_iters) {
    orc::Problem *p = (orc::Problem *)h;
    const int n = p->dimFreeT + p->dimFreeP;
    std::vector<double> x(n);
    if (use_x_as_start) std::copy(x_inout, x_inout + n, x.begin());
    else p->initialGuess(x.data(), current_solver());
    orc_lbfgs::Params pm;                        // CPU.hpp:1243-1247
    pm.mem_size = 128; pm.past = 3; pm.g_epsilon = 1.0e-16; pm.min_step = 1.0e-32; pm.delta = relCostTol;
    pm.max_iterations = max_iterations;          // 0 in the reference (unbounded)
    p->evals = 0;
    double fx = 0.0;
    int ret, iters = 0;
    if (g_hook) ret = hook_solver(n, x.data(), &fx, &orc::Problem::objectiveFunc, p, pm);
    else ret = orc_lbfgs::optimize(n, x.data(), &fx, &orc::Problem::objectiveFunc, p, pm, nullptr, &iters);
    p->finalGenerate(x.data());
    if (x_inout) std::copy(x.begin(), x.end(), x_inout);
    if (C) std::copy(p->jerkOpt.b.begin(), p->jerkOpt.b.end(), C);
    if (T) std::copy(p->fineT.begin(), p->fineT.end(), T);
    if (jerk_cost) *jerk_cost = p->jerkOpt.getTrajJerkCost();
    if (final_obj) *final_obj = fx;
    if (n_evals) *n_evals = p->evals;
    if (n_iters) *n_iters = iters;
    return ret;
}

// record the points evaluated by the next optimize() (lock-step parity tests): cap = max number of points
void orc_trace_begin(void *h, long cap) { orc::Problem *p = (orc::Problem *)h; p->trace.clear(); p->trace_cap = cap; }
long orc_trace_get(void *h, double *out, long cap) {
    orc::Problem *p = (orc::Problem *)h;
    const long n = p->dimFreeT + p->dimFreeP, cnt = std::min<long>((long)(p->trace.size() / (size_t)n), cap);
    if (out) std::copy(p->trace.begin(), p->trace.begin() + cnt * n, out);
    p->trace_cap = 0;
    return cnt;
}

// stand-alone access to the restated L-BFGS (for the three-way solver test): minimises a
// callback objective, optionally recording (fx, step, ls) per iteration
int orc_lbfgs_run(int n, double *x, double *fx, orc_lbfgs::eval_fn fn, void *inst, const double *params11, int trace_cap,
                  double *trace_fx, double *trace_step, int *trace_ls, int *trace_len) {
    orc_lbfgs::Params pm;
    pm.mem_size = (int)params11[0]; pm.g_epsilon = params11[1]; pm.past = (int)params11[2]; pm.delta = params11[3];
    pm.max_iterations = (int)params11[4]; pm.max_linesearch = (int)params11[5]; pm.min_step = params11[6];
    pm.max_step = params11[7]; pm.f_dec_coeff = params11[8]; pm.s_curv_coeff = params11[9]; pm.xtol = params11[10];
    orc_lbfgs::Trace tr;
    int ret = orc_lbfgs::optimize(n, x, fx, fn, inst, pm, nullptr, nullptr, &tr);
    int L = (int)std::min<size_t>(tr.fx.size(), (size_t)trace_cap);
    for (int i = 0; i < L; i++) { trace_fx[i] = tr.fx[i]; trace_step[i] = tr.step[i]; trace_ls[i] = tr.ls[i]; }
    if (trace_len) *trace_len = (int)tr.fx.size();
    return ret;
}
// address of the oracle objective, so an external solver (oracle/_ref) can drive it
void *orc_objective_fnptr() { return (void *)&orc::Problem::objectiveFunc; }

} // extern "C"
