// Host-side neighbours of the hot path (SURVEY.md §8f "next" rows), kept out of the device library's way:
//   f1  H -> V vertex enumeration of the corridor cells and their consecutive overlaps, so that V-polytopes need not be
//       supplied by the caller (reference: SE3GCOPTER::extractVs -> geoutils::enumerateVs, se3gcopter_cpu.hpp:1031-1074,
//       geoutils.hpp:43-149: Seidel LP interior point + polar-dual quickhull + quantised de-duplication);
//   f3  the result wire format: Trajectory -> quadrotor_msgs/PolynomialTrajectory fields (MavGlobalPlanner::traj2msg,
//       se3_planner.cpp:31-58, with Piece::normalizePosCoeffMat, trajectory.hpp:131-141) and the consumer's sampling of it
//       (traj_server.cpp:406-456).
// f1 does NOT reproduce the reference's vertex ORDER: that order falls out of an LP whose constraint permutation comes
// from a process-global RNG (sdlp.hpp:689-708) and of quickhull's facet order.  Any fixed order
// is a valid parameterisation of the same polytope (the xi -> q map of se3gcopter_cpu.hpp:729-747 is onto the polytope for
// every vertex order); here vertices are sorted lexicographically so that the order is a function of the polytope alone.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/frx.h"

namespace {

// all vertices of { x : n_k . (x - p_k) <= 0 } by intersecting every triple of planes; K <= a few dozen
int enumerate(int K, const double *h, std::vector<std::array<double, 3>> &out, double tol = 1e-9, double quant = 1e-7) {
    std::vector<std::array<double, 4>> pl(K);                 // unit normal, offset d: n.x <= d
    for (int k = 0; k < K; k++) {
        const double *r = h + 6 * k;
        const double nn = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        pl[k] = {r[0] / nn, r[1] / nn, r[2] / nn, (r[0] * r[3] + r[1] * r[4] + r[2] * r[5]) / nn};
    }
    std::vector<std::array<double, 3>> vs;
    for (int a = 0; a < K; a++)
        for (int b = a + 1; b < K; b++)
            for (int c = b + 1; c < K; c++) {
                const double *A = pl[a].data(), *B = pl[b].data(), *C = pl[c].data();
                const double cx = B[1] * C[2] - B[2] * C[1], cy = B[2] * C[0] - B[0] * C[2], cz = B[0] * C[1] - B[1] * C[0];   // B x C
                const double det = A[0] * cx + A[1] * cy + A[2] * cz;
                if (std::fabs(det) <= 1e-10) continue;
                // x = (d_a (B x C) + d_b (C x A) + d_c (A x B)) / det
                const double ax = C[1] * A[2] - C[2] * A[1], ay = C[2] * A[0] - C[0] * A[2], az = C[0] * A[1] - C[1] * A[0];   // C x A
                const double bx = A[1] * B[2] - A[2] * B[1], by = A[2] * B[0] - A[0] * B[2], bz = A[0] * B[1] - A[1] * B[0];   // A x B
                const std::array<double, 3> x = {(A[3] * cx + B[3] * ax + C[3] * bx) / det, (A[3] * cy + B[3] * ay + C[3] * by) / det,
                                                 (A[3] * cz + B[3] * az + C[3] * bz) / det};
                bool feas = true;
                for (int k = 0; k < K && feas; k++) feas = pl[k][0] * x[0] + pl[k][1] * x[1] + pl[k][2] * x[2] <= pl[k][3] + tol;
                if (feas) vs.push_back(x);
            }
    // de-duplicate on a quant grid (first occurrence wins), then sort lexicographically
    // (ordering by grid key rather than by coordinate keeps ties such as x = -7.6343.. +- 1 ulp from flipping the order)
    using Key = std::array<long long, 3>;
    std::vector<std::pair<Key, std::array<double, 3>>> uniq;
    for (const auto &v : vs) {
        const Key key = {(long long)std::nearbyint(v[0] / quant), (long long)std::nearbyint(v[1] / quant), (long long)std::nearbyint(v[2] / quant)};
        bool seen = false;
        for (const auto &u : uniq) seen = seen || u.first == key;
        if (!seen) uniq.push_back({key, v});
    }
    std::sort(uniq.begin(), uniq.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    out.clear();
    for (const auto &u : uniq) out.push_back(u.second);
    return (int)out.size();
}

} // namespace

extern "C" {

int frx_enumerate_vertices(int K, const double *h_rec, double *v_out, int cap, int *nv) {
    if (K < 4 || !h_rec || !nv) return FRX_ERR_INVALID_ARG;
    std::vector<std::array<double, 3>> vs;
    *nv = enumerate(K, h_rec, vs);
    if (*nv < 4) return FRX_ERR_EMPTY_POLYTOPE;                 // no interior (setup() returns false, se3gcopter_cpu.hpp:1118-1121)
    if (v_out) {
        if (cap < *nv) return FRX_ERR_CAPACITY;
        for (int i = 0; i < *nv; i++) std::memcpy(v_out + 3 * i, vs[i].data(), sizeof(double) * 3);
    }
    return FRX_OK;
}

int frx_problem_create_from_h(const frx_config *cfg, int device, int B, const int *coarse_n, const double *ini_state,
                              const double *fin_state, const int *h_off, const double *h_rec, frx_problem **out) {
    if (!cfg || !coarse_n || !h_off || !h_rec || !out || B <= 0) return FRX_ERR_INVALID_ARG;
    std::vector<int> v_off{0};
    std::vector<double> v_rec;
    std::vector<std::array<double, 3>> vs;
    std::vector<double> both;
    int poly = 0;
    for (int b = 0; b < B; b++) {
        for (int i = 0; i < coarse_n[b]; i++) {
            const int hb = h_off[poly + i], K = h_off[poly + i + 1] - hb;
            if (enumerate(K, h_rec + 6 * (size_t)hb, vs) < 4) return FRX_ERR_EMPTY_POLYTOPE;
            v_off.push_back(v_off.back() + (int)vs.size());
            for (const auto &v : vs) v_rec.insert(v_rec.end(), v.begin(), v.end());
            if (i + 1 < coarse_n[b]) {                               // overlap of consecutive cells (se3gcopter_cpu.hpp:1052-1054)
                const int K2 = h_off[poly + i + 2] - h_off[poly + i + 1];
                both.assign(h_rec + 6 * (size_t)hb, h_rec + 6 * (size_t)(hb + K + K2));
                if (enumerate(K + K2, both.data(), vs) < 4) return FRX_ERR_EMPTY_POLYTOPE;
                v_off.push_back(v_off.back() + (int)vs.size());
                for (const auto &v : vs) v_rec.insert(v_rec.end(), v.begin(), v.end());
            }
        }
        poly += coarse_n[b];
    }
    return frx_problem_create(cfg, device, B, coarse_n, ini_state, fin_state, h_off, h_rec, v_off.data(), v_rec.data(), out);
}

int frx_traj_to_msg(int n_pieces, const double *T, const double *C, double *coef_x, double *coef_y, double *coef_z, double *time,
                    unsigned *order) {
    if (n_pieces <= 0 || !T || !C || !coef_x || !coef_y || !coef_z || !time || !order) return FRX_ERR_INVALID_ARG;
    for (int i = 0; i < n_pieces; i++) {
        // Piece holds the 3x6 matrix highest power first (getTraj, se3gcopter_cpu.hpp:561); normalizePosCoeffMat scales column j
        // (power 5-j) by duration^(5-j) (trajectory.hpp:131-141); traj2msg pushes the columns in that order (se3_planner.cpp:44-51)
        double tp = 1.0;
        for (int j = 5; j >= 0; j--) {                                  // column j <-> power k = 5 - j
            const int k = 5 - j;
            coef_x[6 * i + j] = C[18 * (size_t)i + 3 * k + 0] * tp;
            coef_y[6 * i + j] = C[18 * (size_t)i + 3 * k + 1] * tp;
            coef_z[6 * i + j] = C[18 * (size_t)i + 3 * k + 2] * tp;
            tp *= T[i];
        }
        time[i] = T[i];
        order[i] = 5;
    }
    return FRX_OK;
}

int frx_msg_sample(int n_segment, const double *coef_x, const double *coef_y, const double *coef_z, const double *time,
                   const unsigned *order, double t, double *pos, double *vel, double *acc, double *jerk) {
    if (n_segment <= 0 || !coef_x || !coef_y || !coef_z || !time || !order || !pos || !vel || !acc || !jerk) return FRX_ERR_INVALID_ARG;
    // traj_server.cpp:406-456
    t = std::max(0.0, t);
    int seg = 0, shift = 0;
    double dur = 0.0;
    for (seg = 0; seg < n_segment && t > (dur = time[seg]); seg++) { t -= dur; shift += (int)order[seg] + 1; }
    if (seg == n_segment) { seg--; shift -= (int)order[seg] + 1; t += time[seg]; }
    t /= time[seg];
    const int cur_order = (int)order[seg];
    const double *cf[3] = {coef_x + shift, coef_y + shift, coef_z + shift};
    for (int a = 0; a < 3; a++) {
        double p = 0.0, v = 0.0, ac = 0.0, jk = 0.0;
        double tn = 1.0, tnvel = 1.0, tnacc = 1.0, tnjerk = 1.0;
        int n = 1, k = 1, l = 2, j1 = 1, j2 = 2, j3 = 3;
        for (int i = cur_order; i >= 0; i--) {
            p += tn * cf[a][i];
            tn *= t;
            if (i <= cur_order - 1) {
                v += n * tnvel * cf[a][i]; tnvel *= t; n++;
                if (i <= cur_order - 2) {
                    ac += l * k * tnacc * cf[a][i]; tnacc *= t; l++; k++;
                    if (i <= cur_order - 3) { jk += j1 * j2 * j3 * tnjerk * cf[a][i]; tnjerk *= t; j1++; j2++; j3++; }
                }
            }
        }
        pos[a] = p; vel[a] = v / time[seg]; acc[a] = ac / (time[seg] * time[seg]); jerk[a] = jk / (time[seg] * time[seg] * time[seg]);
    }
    return FRX_OK;
}

} // extern "C"
