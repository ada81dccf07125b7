// The dense state of the compact L-BFGS representation (Byrd-Nocedal-Schnabel; frx_round_kernel.hpp, rk_dense_loop) rebuilt from a history of pairs -
// host code.  The resident round kernel maintains R^-1, Y^T Y and D = diag(s_i . y_i) incrementally, one column per accepted step; a plan that
// the per-stage rounds began (their history lives in HBM rows, frx_lbfgs_kernels.hpp) and that continues on the resident kernel needs the same
// three objects for the pairs it already has.  Same pairs, same matrices: the direction that follows is the two-loop recursion's, to rounding.
#pragma once
#include <cstddef>
#include <vector>

namespace frx {

// S, Y: [m][hs] rows by SLOT, natural element order, n <= hs significant doubles per row (zero beyond); the `bound` newest pairs are valid, the newest
// in slot `newest`, older ones in newest - 1, newest - 2, ... (mod m).
// rinv [128][rs]: R^-1 indexed by slot, R_ij = s_i . y_j for pair i not newer than pair j (upper triangular in age order), zero elsewhere;
// yy [128][128]: y_i . y_j for valid slots, zero elsewhere;  vd [128]: s_i . y_i.
inline void compact_from_history(int m, int n, size_t hs, int bound, int newest, const double *S, const double *Y, int rs, double *rinv, double *yy, double *vd) {
    for (size_t i = 0; i < (size_t)128 * rs; i++) rinv[i] = 0.0;
    for (size_t i = 0; i < (size_t)128 * 128; i++) yy[i] = 0.0;
    for (int i = 0; i < 128; i++) vd[i] = 0.0;
    if (bound <= 0) return;
    const int b = bound;
    std::vector<int> slot(b);                                  // age order: index 0 = oldest ... b - 1 = newest
    for (int a = 0; a < b; a++) { int j = newest - (b - 1 - a); while (j < 0) j += m; slot[a] = j % m; }
    std::vector<double> R((size_t)b * b, 0.0), Ri((size_t)b * b, 0.0);
    auto dot = [n](const double *u, const double *w) { double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0; int i = 0; for (; i + 4 <= n; i += 4) { s0 += u[i] * w[i]; s1 += u[i + 1] * w[i + 1]; s2 += u[i + 2] * w[i + 2]; s3 += u[i + 3] * w[i + 3]; } for (; i < n; i++) s0 += u[i] * w[i]; return (s0 + s1) + (s2 + s3); };
    for (int a = 0; a < b; a++) {
        const double *sa = S + (size_t)slot[a] * hs, *ya = Y + (size_t)slot[a] * hs;
        for (int c = a; c < b; c++) {
            const double *yc = Y + (size_t)slot[c] * hs;
            R[(size_t)a * b + c] = dot(sa, yc);
            const double q = dot(ya, yc);
            yy[(size_t)slot[a] * 128 + slot[c]] = q; yy[(size_t)slot[c] * 128 + slot[a]] = q;
        }
        vd[slot[a]] = R[(size_t)a * b + a];
    }
    // inverse of the upper triangular R, column by column: R Ri = I
    for (int c = 0; c < b; c++) {
        Ri[(size_t)c * b + c] = 1.0 / R[(size_t)c * b + c];
        for (int a = c - 1; a >= 0; a--) {
            double acc = 0.0;
            for (int k = a + 1; k <= c; k++) acc += R[(size_t)a * b + k] * Ri[(size_t)k * b + c];
            Ri[(size_t)a * b + c] = -acc / R[(size_t)a * b + a];
        }
    }
    for (int a = 0; a < b; a++)
        for (int c = a; c < b; c++) rinv[(size_t)slot[a] * rs + slot[c]] = Ri[(size_t)a * b + c];
}

} // namespace frx
