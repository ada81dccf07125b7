// Host-side neighbours of the hot path (SURVEY.md §8f "next" rows), kept out of the device library's way:
//   f1  H -> V vertex enumeration of the corridor cells and their consecutive overlaps, so that V-polytopes need not be
//       supplied by the caller (reference: SE3GCOPTER::extractVs -> geoutils::enumerateVs, se3gcopter_cpu.hpp:1031-1074,
//       geoutils.hpp:43-149: Seidel LP interior point + polar-dual quickhull + quantised de-duplication);
//   f2  safe-flight-corridor generation: a line segment of the front-end path is inflated into the largest obstacle-free
//       ellipsoid, obstacle points become tangent half-spaces, a local bounding box closes the cell (decomp_util:
//       line_segment.h:31-35,47-85,136-214, decomp_base.h:35-83, ellipsoid.h:19-61), and cells are chained greedily along the
//       path (MavGlobalPlanner::plan, MinCoPlan_CPU.cpp:37-105);
//   f3  the result wire format: Trajectory -> quadrotor_msgs/PolynomialTrajectory fields (MavGlobalPlanner::traj2msg,
//       se3_planner.cpp:31-58, with Piece::normalizePosCoeffMat, trajectory.hpp:131-141) and the consumer's sampling of it
//       (traj_server.cpp:406-456).
// f1 does NOT reproduce the reference's vertex ORDER: that order falls out of an LP whose constraint permutation comes
// from a process-global RNG (sdlp.hpp:689-708) and of quickhull's facet order.  Any fixed order
// is a valid parameterisation of the same polytope (the xi -> q map of se3gcopter_cpu.hpp:729-747 is onto the polytope for
// every vertex order); here vertices are sorted lexicographically so that the order is a function of the polytope alone.
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/frx.h"
#include "frx_internal.hpp"
#include "frx_host_pool.hpp"

namespace {

enum { POLY_OK = 0, POLY_UNBOUNDED = 1, POLY_FLAT = 2 };

// all vertices of { x : n_k . (x - p_k) <= 0 } by intersecting every triple of planes; K <= a few dozen
// `verdict` (optional): what geoutils::findInterior decides before the reference enumerates anything (geoutils.hpp:43-77: the
// Chebyshev-centre LP must be bounded with a strictly positive radius, else enumerateVs and with it SE3GCOPTER::setup return
// false, se3gcopter_cpu.hpp:1118-1121).  Here: POLY_UNBOUNDED when the recession cone {u : n_k . u <= 0} is not {0} (normals
// of rank < 3, or an extreme ray n_a x n_b of the cone survives every plane), POLY_FLAT when the vertex centroid - strictly
// interior for a full-dimensional polytope - has no positive slack on some plane (zero volume: e.g. two cells sharing a face).
int enumerate(int K, const double *h, std::vector<std::array<double, 3>> &out, int *verdict = nullptr, double tol = 1e-9, double quant = 1e-7) {
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
    if (verdict) {
        *verdict = POLY_OK;
        // boundedness: every extreme ray of a pointed recession cone lies along n_a x n_b for some pair of planes; a cone with a
        // lineality space means the normals do not span R^3, in which case no plane triple had a non-zero determinant above
        bool spans = false, ray = false;
        for (int a = 0; a < K && !ray; a++)
            for (int b = a + 1; b < K && !ray; b++) {
                const double *A = pl[a].data(), *B = pl[b].data();
                double u[3] = {A[1] * B[2] - A[2] * B[1], A[2] * B[0] - A[0] * B[2], A[0] * B[1] - A[1] * B[0]};
                const double un = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
                if (un <= 1e-10) continue;
                for (double &c : u) c /= un;
                for (int k = 0; k < K && !spans; k++) spans = std::fabs(pl[k][0] * u[0] + pl[k][1] * u[1] + pl[k][2] * u[2]) > 1e-10;
                for (int sgn = -1; sgn <= 1 && !ray; sgn += 2) {
                    bool free_dir = true;
                    for (int k = 0; k < K && free_dir; k++) free_dir = sgn * (pl[k][0] * u[0] + pl[k][1] * u[1] + pl[k][2] * u[2]) <= 1e-12;
                    ray = free_dir;
                }
            }
        if (!spans || ray) *verdict = POLY_UNBOUNDED;
        else if (out.size() < 4) *verdict = POLY_FLAT;
        else {
            double c[3] = {0.0, 0.0, 0.0};
            for (const auto &v : out) { c[0] += v[0]; c[1] += v[1]; c[2] += v[2]; }
            for (double &q : c) q /= (double)out.size();
            double slack = std::numeric_limits<double>::max();
            for (int k = 0; k < K; k++) slack = std::min(slack, pl[k][3] - (pl[k][0] * c[0] + pl[k][1] * c[1] + pl[k][2] * c[2]));
            if (!(slack > tol)) *verdict = POLY_FLAT;
        }
    }
    return (int)out.size();
}

int polytope_error(int verdict, const char *what, int index) {
    return frx::set_error(FRX_ERR_EMPTY_POLYTOPE, std::string(what) + " " + std::to_string(index) +
                          (verdict == POLY_UNBOUNDED ? " is unbounded" : " has no interior (zero volume)") +
                          ": SE3GCOPTER::setup returns false here (geoutils::findInterior, geoutils.hpp:43-77)");
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// f2: corridor cells
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr double kDecompEps = 1e-10;                               // decomp_basis/data_type.h:129

struct V3 { double x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
struct M3 { double m[3][3]; };
inline V3 mul(const M3 &A, V3 v) { return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z, A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z}; }
inline V3 mulT(const M3 &A, V3 v) { return {A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z, A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z, A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z}; }
inline M3 mul(const M3 &A, const M3 &B) {
    M3 C;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}
inline M3 transpose(const M3 &A) { M3 T; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T.m[i][j] = A.m[j][i]; return T; }
inline M3 inverse(const M3 &A) {
    auto cof = [&](int i, int j) { const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return A.m[i1][j1] * A.m[i2][j2] - A.m[i1][j2] * A.m[i2][j1]; };
    const double det = cof(0, 0) * A.m[0][0] + cof(1, 0) * A.m[1][0] + cof(2, 0) * A.m[2][0];
    M3 I;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I.m[j][i] = cof(i, j) / det;
    return I;
}
// rotation that takes e_x to the direction of v with zero roll: R = Rz(yaw) Ry(pitch)   (geometric_utils.h:27-35)
inline M3 rotation_from_direction(V3 v) {
    const double pitch = std::atan2(-v.z, std::hypot(v.x, v.y)), yaw = std::atan2(v.y, v.x);
    const double cp = std::cos(pitch), sp = std::sin(pitch), cy = std::cos(yaw), sy = std::sin(yaw);
    return M3{{{cy * cp, -sy, cy * sp}, {sy * cp, cy, sy * sp}, {-sp, 0.0, cp}}};
}
inline M3 roll_about_x(double roll) { const double c = std::cos(roll), s = std::sin(roll); return M3{{{1.0, 0.0, 0.0}, {0.0, c, -s}, {0.0, s, c}}}; }

struct Plane { V3 n, p; };                                         // outside is n . (x - p) > 0
struct Ellipsoid3 {
    M3 C, Cinv; V3 d;                                              // { C u + d : |u| <= 1 }
    void set_shape(const M3 &R, double a0, double a1, double a2) {
        const M3 D{{{a0, 0.0, 0.0}, {0.0, a1, 0.0}, {0.0, 0.0, a2}}};
        C = mul(mul(R, D), transpose(R));
        Cinv = inverse(C);
    }
    double dist(V3 pt) const { return norm(mul(Cinv, pt - d)); }   // ellipsoid.h:19-21
    int closest(const std::vector<V3> &pts) const {                // ellipsoid.h:39-50: first of the minima
        int best = -1; double dmin = std::numeric_limits<double>::max();
        for (size_t i = 0; i < pts.size(); i++) { const double di = dist(pts[i]); if (di < dmin) { dmin = di; best = (int)i; } }
        return best;
    }
};

void local_bbox_planes(V3 p1, V3 p2, const double *bbox, std::vector<Plane> &out) {          // line_segment.h:47-85
    if (std::sqrt(bbox[0] * bbox[0] + bbox[1] * bbox[1] + bbox[2] * bbox[2]) == 0.0) return;
    V3 dir = p2 - p1; dir = dir * (1.0 / norm(dir));
    V3 h{dir.y, -dir.x, 0.0};
    if (norm(h) == 0.0) h = V3{-1.0, 0.0, 0.0};
    h = h * (1.0 / norm(h));
    out.push_back({h, p1 + h * bbox[1]}); out.push_back({h * -1.0, p1 - h * bbox[1]});
    out.push_back({dir, p2 + dir * bbox[0]}); out.push_back({dir * -1.0, p1 - dir * bbox[0]});
    const V3 v{dir.y * h.z - dir.z * h.y, dir.z * h.x - dir.x * h.z, dir.x * h.y - dir.y * h.x};
    out.push_back({v, p1 + v * bbox[2]}); out.push_back({v * -1.0, p1 - v * bbox[2]});
}
inline bool inside_planes(const std::vector<Plane> &pl, V3 pt) {    // polyhedron.h: Polyhedron::inside
    for (const auto &h : pl) if (dot(h.n, pt - h.p) > kDecompEps) return false;
    return true;
}

// LineSegment3D::dilate (line_segment.h:31-35): ellipsoid, tangent half-spaces, local bounding box
void dilate_segment(V3 p1, V3 p2, const double *bbox, const std::vector<V3> &cloud, double offset, std::vector<Plane> &planes, Ellipsoid3 &E) {
    std::vector<Plane> box;
    local_bbox_planes(p1, p2, bbox, box);
    std::vector<V3> obs;                                           // decomp_base.h:35-40: only the points inside the local box count
    for (const V3 &q : cloud) if (inside_planes(box, q)) obs.push_back(q);

    // --- find_ellipsoid (line_segment.h:136-214)
    const double f = norm(p1 - p2) / 2;
    double a0 = f + offset, a1 = f, a2 = f;
    if (a0 > 0) { const double ratio = a1 / a0; a0 *= ratio; a1 *= ratio; a2 *= ratio; }       // (C scaled alike)
    const M3 Ri = rotation_from_direction(p2 - p1);
    E.d = (p1 + p2) * 0.5;
    E.set_shape(Ri, a0, a1, a2);
    std::vector<V3> in0;
    for (const V3 &q : obs) if (E.dist(q) <= 1) in0.push_back(q);
    M3 Rf = Ri;
    std::vector<V3> live = in0, next;
    while (!live.empty()) {                                        // shrink the two short axes together, rolled towards the closest point
        const V3 pw = live[E.closest(live)];
        V3 p = mulT(Ri, pw - E.d);
        Rf = mul(Ri, roll_about_x(std::atan2(p.z, p.y)));
        p = mulT(Rf, pw - E.d);
        if (p.x < a0) a1 = std::fabs(p.y) / std::sqrt(1 - (p.x / a0) * (p.x / a0));
        E.set_shape(Rf, a0, a1, a1);
        next.clear();
        for (const V3 &q : live) if (1 - E.dist(q) > kDecompEps) next.push_back(q);
        live.swap(next);
    }
    E.set_shape(Rf, a0, a1, a2);                                   // the third axis starts again from its old length
    live.clear();
    for (const V3 &q : in0) if (E.dist(q) <= 1) live.push_back(q);
    while (!live.empty()) {
        const V3 pw = live[E.closest(live)];
        const V3 p = mulT(Rf, pw - E.d);
        const double dd = 1 - (p.x / a0) * (p.x / a0) - (p.y / a1) * (p.y / a1);
        if (dd > kDecompEps) a2 = std::fabs(p.z) / std::sqrt(dd);
        E.set_shape(Rf, a0, a1, a2);
        next.clear();
        for (const V3 &q : live) if (1 - E.dist(q) > kDecompEps) next.push_back(q);
        live.swap(next);
    }
    // --- find_polyhedron (decomp_base.h:63-83): the tangent plane at the closest point cuts away everything behind it
    planes.clear();
    live = obs;
    const M3 Q = mul(E.Cinv, transpose(E.Cinv));
    while (!live.empty()) {
        const V3 c = live[E.closest(live)];
        V3 n = mul(Q, c - E.d); n = n * (1.0 / norm(n));           // ellipsoid.h:53-58
        planes.push_back({n, c});
        next.clear();
        for (const V3 &q : live) if (dot(n, q - c) < 0) next.push_back(q);
        live.swap(next);
    }
    planes.insert(planes.end(), box.begin(), box.end());
}

} // namespace

extern "C" {

int frx_line_segment_dilate(const double *p1, const double *p2, const double *bbox, int n_obs, const double *obs, double offset, int cap,
                            int *n_planes, double *h_rec, double *ell_C, double *ell_d) {
    if (!p1 || !p2 || !bbox || n_obs < 0 || (n_obs && !obs) || !n_planes) return frx::set_error(FRX_ERR_INVALID_ARG, "frx_line_segment_dilate: null or out-of-range argument");
    std::vector<V3> cloud(n_obs);
    for (int i = 0; i < n_obs; i++) cloud[i] = V3{obs[3 * i], obs[3 * i + 1], obs[3 * i + 2]};
    std::vector<Plane> pl; Ellipsoid3 E;
    dilate_segment(V3{p1[0], p1[1], p1[2]}, V3{p2[0], p2[1], p2[2]}, bbox, cloud, offset, pl, E);
    *n_planes = (int)pl.size();
    if (h_rec) {
        if (cap < (int)pl.size()) return frx::set_error(FRX_ERR_CAPACITY, "frx_line_segment_dilate: output capacity too small");
        for (size_t k = 0; k < pl.size(); k++) { double *r = h_rec + 6 * k; r[0] = pl[k].n.x; r[1] = pl[k].n.y; r[2] = pl[k].n.z; r[3] = pl[k].p.x; r[4] = pl[k].p.y; r[5] = pl[k].p.z; }
    }
    if (ell_C) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) ell_C[3 * i + j] = E.C.m[i][j];
    if (ell_d) { ell_d[0] = E.d.x; ell_d[1] = E.d.y; ell_d[2] = E.d.z; }
    return FRX_OK;
}

int frx_corridor_generate(int n_path, const double *path, int n_obs, const double *obs, const double *bbox, double map_height, double max_seg,
                          frx_blocked_fn blocked, void *user, int cap_polys, int cap_planes, int *n_polys, int *h_off, double *h_rec) {
    if (n_path < 2 || !path || !bbox || n_obs < 0 || (n_obs && !obs) || !n_polys || !h_off || !h_rec || cap_polys < 1) return frx::set_error(FRX_ERR_INVALID_ARG, "frx_corridor_generate: null or out-of-range argument");
    std::vector<V3> cloud(n_obs);
    for (int i = 0; i < n_obs; i++) cloud[i] = V3{obs[3 * i], obs[3 * i + 1], obs[3 * i + 2]};
    auto P = [&](int i) { return V3{path[3 * i], path[3 * i + 1], path[3 * i + 2]}; };
    std::vector<Plane> pl; Ellipsoid3 E;
    int np = 0, used = 0;
    h_off[0] = 0;
    for (int i = 0; i < n_path - 1;) {                             // MinCoPlan_CPU.cpp:44-83
        int k;
        for (k = i + 1; k < n_path; k++) {                         // the farthest point still visible from path[i] and closer than max_seg
            const bool hit = blocked ? blocked(path + 3 * i, path + 3 * k, user) != 0 : false;
            if (hit || norm(P(i) - P(k)) >= max_seg) { k--; break; }
        }
        if (k < i + 1) k = i + 1;
        if (k >= n_path) k = n_path - 1;
        dilate_segment(P(i), P(k), bbox, cloud, 0.0, pl, E);
        int j;
        for (j = k; j < n_path; j++) if (!inside_planes(pl, P(j))) break;    // how far the path stays inside this cell (tested before
        j--;                                                                 // the floor / ceiling planes are added, as in the reference)
        pl.push_back({V3{0.0, 0.0, 1.0}, V3{0.0, 0.0, map_height}});         // MinCoPlan_CPU.cpp:85-91
        pl.push_back({V3{0.0, 0.0, -1.0}, V3{0.0, 0.0, 0.0}});
        if (np >= cap_polys || used + (int)pl.size() > cap_planes) return frx::set_error(FRX_ERR_CAPACITY, "frx_corridor_generate: output capacity too small");
        for (const Plane &h : pl) { double *r = h_rec + 6 * (size_t)used++; r[0] = h.n.x; r[1] = h.n.y; r[2] = h.n.z; r[3] = h.p.x; r[4] = h.p.y; r[5] = h.p.z; }
        h_off[++np] = used;
        if (j >= n_path - 1) break;
        const int wp = (1 * i + 4 * j) / 5;                        // restart 4/5 of the way to where the path leaves the cell
        i = wp > i ? wp : i + 1;                                   // (the reference would not advance, i.e. hang, when wp == i)
    }
    *n_polys = np;
    return FRX_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// f3: post-checks — largest speed and largest acceleration magnitude of every piece (Piece::getMaxVelRate / getMaxAccRate,
// trajectory.hpp:177-273).  The reference normalises time to [0, 1], forms d/dtau |v|^2 (degree 7; degree 5 for |a|^2) and
// isolates its roots with Sturm sequences (root_finder.hpp); here the roots are bracketed by the roots of the derivative,
// recursively from degree 1 upwards, and refined by bisection — every sign change of the polynomial on [0, 1] is found, which is
// all a maximum needs (a root of even multiplicity is an inflection of |v|^2, not an extremum).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

inline double horner(const double *c, int deg, double x) {          // c[0] x^deg + ... + c[deg]
    double v = c[0];
    for (int i = 1; i <= deg; i++) v = v * x + c[i];
    return v;
}
// all sign-change roots of c (highest power first, degree deg) in [0, 1], ascending
void roots_unit(const double *c, int deg, std::vector<double> &out) {
    out.clear();
    while (deg > 0 && c[0] == 0.0) { c++; deg--; }
    if (deg <= 0) return;
    if (deg == 1) { const double r = -c[1] / c[0]; if (r >= 0.0 && r <= 1.0) out.push_back(r); return; }
    double dc[8];
    for (int i = 0; i < deg; i++) dc[i] = c[i] * (deg - i);
    std::vector<double> crit;
    roots_unit(dc, deg - 1, crit);
    std::vector<double> pts{0.0};
    pts.insert(pts.end(), crit.begin(), crit.end());
    pts.push_back(1.0);
    for (size_t i = 0; i + 1 < pts.size(); i++) {
        double a = pts[i], b = pts[i + 1], fa = horner(c, deg, a), fb = horner(c, deg, b);
        if (fa == 0.0) { if (out.empty() || out.back() != a) out.push_back(a); continue; }
        if (fb == 0.0) { out.push_back(b); continue; }
        if ((fa < 0.0) == (fb < 0.0)) continue;
        for (int it = 0; it < 200 && b - a > 0.0; it++) {            // monotone on (a, b): plain bisection down to adjacent doubles
            const double m = 0.5 * (a + b);
            if (m <= a || m >= b) break;
            const double fm = horner(c, deg, m);
            if (fm == 0.0) { a = b = m; break; }
            if ((fm < 0.0) == (fa < 0.0)) { a = m; fa = fm; } else { b = m; }
        }
        out.push_back(0.5 * (a + b));
    }
}
// max over tau in [0,1] of | sum_k w[k][.] tau^k |^2 for a vector polynomial of degree `deg` (w[k][d], lowest power first)
double max_sq_norm(const double (*w)[3], int deg) {
    double sq[16] = {0};                                             // |w|^2, lowest power first, degree 2 deg
    for (int i = 0; i <= deg; i++) for (int j = 0; j <= deg; j++) sq[i + j] += w[i][0] * w[j][0] + w[i][1] * w[j][1] + w[i][2] * w[j][2];
    const int d2 = 2 * deg;
    double der[16];                                                  // derivative, highest power first
    for (int k = d2; k >= 1; k--) der[d2 - k] = k * sq[k];
    double dn = 0.0;
    for (int i = 0; i < d2; i++) dn += der[i] * der[i];
    if (dn < 2.220446049250313e-16) return 0.0;                      // the reference reports 0 for a (numerically) constant magnitude,
    std::vector<double> cand;                                        // zero or not (trajectory.hpp:192-195, 240-243): kept
    roots_unit(der, d2 - 1, cand);
    cand.push_back(0.0); cand.push_back(1.0);
    double best = 0.0;
    for (double t : cand) {
        double v[3] = {0, 0, 0}, tn = 1.0;
        for (int k = 0; k <= deg; k++) { v[0] += w[k][0] * tn; v[1] += w[k][1] * tn; v[2] += w[k][2] * tn; tn *= t; }
        best = std::max(best, v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    }
    return best;
}

} // namespace

extern "C" {

int frx_traj_max_rates(int n_pieces, const double *T, const double *C, double *max_vel, double *max_acc) {
    if (n_pieces <= 0 || !T || !C || (!max_vel && !max_acc)) return frx::set_error(FRX_ERR_INVALID_ARG, "frx_traj_max_rates: null or out-of-range argument");
    for (int i = 0; i < n_pieces; i++) {
        const double *c = C + 18 * (size_t)i, h = T[i];
        // derivatives with respect to normalised time tau = t / h, lowest power first, scaled as normalizeVelCoeffMat /
        // normalizeAccCoeffMat scale them (trajectory.hpp:147-175): dp/dtau = h v, d2p/dtau2 = h^2 a
        double wv[5][3], wa[4][3], hp = h;
        for (int k = 0; k <= 4; k++) { for (int d = 0; d < 3; d++) wv[k][d] = (k + 1) * c[3 * (k + 1) + d] * hp; hp *= h; }
        hp = h * h;
        for (int k = 0; k <= 3; k++) { for (int d = 0; d < 3; d++) wa[k][d] = (k + 2) * (k + 1) * c[3 * (k + 2) + d] * hp; hp *= h; }
        if (max_vel) max_vel[i] = std::sqrt(max_sq_norm(wv, 4)) / h;
        if (max_acc) max_acc[i] = std::sqrt(max_sq_norm(wa, 3)) / (h * h);
    }
    return FRX_OK;
}

} // extern "C"

extern "C" {

int frx_enumerate_vertices(int K, const double *h_rec, double *v_out, int cap, int *nv) {
    if (K < 4 || !h_rec || !nv) return frx::set_error(FRX_ERR_INVALID_ARG, "frx_enumerate_vertices: K < 4 or null argument");
    std::vector<std::array<double, 3>> vs;
    int verdict = POLY_OK;
    *nv = enumerate(K, h_rec, vs, &verdict);
    if (verdict != POLY_OK) return polytope_error(verdict, "polytope", 0);   // setup() returns false, se3gcopter_cpu.hpp:1118-1121
    if (v_out) {
        if (cap < *nv) return frx::set_error(FRX_ERR_CAPACITY, "frx_enumerate_vertices: output capacity too small");
        for (int i = 0; i < *nv; i++) std::memcpy(v_out + 3 * i, vs[i].data(), sizeof(double) * 3);
    }
    return FRX_OK;
}

int frx_problem_create_from_h(const frx_config *cfg, int device, int B, const int *coarse_n, const double *ini_state,
                              const double *fin_state, const int *h_off, const double *h_rec, frx_problem **out) {
    if (!cfg || !coarse_n || !h_off || !h_rec || !out || B <= 0) return frx::set_error(FRX_ERR_INVALID_ARG, "frx_problem_create_from_h: null argument or B <= 0");
    // extractVs (se3gcopter_cpu.hpp:1031-1074) enumerates every cell and every overlap of consecutive cells, one after the other, inside the
    // reference's plan timer.  The 2 cN - 1 polytopes of every candidate are independent: one flat task list for the thread pool, results
    // concatenated in polytope order afterwards (the output does not depend on the number of threads).
    struct Task { int hb, K, K2, poly; };                                // K2 > 0: overlap of cell `poly` with the next one
    std::vector<Task> tasks;
    int poly = 0;
    for (int b = 0; b < B; b++) {
        for (int i = 0; i < coarse_n[b]; i++) {
            const int hb = h_off[poly + i], K = h_off[poly + i + 1] - hb;
            tasks.push_back({hb, K, 0, poly + i});
            if (i + 1 < coarse_n[b]) tasks.push_back({hb, K, h_off[poly + i + 2] - h_off[poly + i + 1], poly + i});
        }
        poly += coarse_n[b];
    }
    const int NT = (int)tasks.size();
    std::vector<std::vector<std::array<double, 3>>> res(NT);
    std::vector<int> verdicts(NT, POLY_OK);
    auto work = [&](int t) {
        const Task &k = tasks[t];
        enumerate(k.K + k.K2, h_rec + 6 * (size_t)k.hb, res[t], &verdicts[t]);    // (an overlap's records are the two cells' records, contiguous in h_rec)
    };
    const auto tdbg0 = std::chrono::steady_clock::now();
    const int nthr = frx::setup_threads(NT, 4.0);                                                // ~4 us per polytope (8 ... 16 planes)
    frx::TaskPool::get().run(NT, nthr, [&](int t, int) { work(t); });
    if (std::getenv("FRX_SETUP_TIMING")) fprintf(stderr, "[frx setup] H->V enumeration: %d polytopes on %d threads, %.3f ms\n", NT, nthr, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tdbg0).count());
    std::vector<int> v_off{0};
    std::vector<double> v_rec;
    for (int t = 0; t < NT; t++) {                                          // first failing polytope in order, as the serial walk reported it
        if (verdicts[t] != POLY_OK) return polytope_error(verdicts[t], tasks[t].K2 ? "overlap of corridor cells" : "corridor cell", tasks[t].poly);
        v_off.push_back(v_off.back() + (int)res[t].size());
        for (const auto &v : res[t]) v_rec.insert(v_rec.end(), v.begin(), v.end());
    }
    return frx_problem_create(cfg, device, B, coarse_n, ini_state, fin_state, h_off, h_rec, v_off.data(), v_rec.data(), out);
}

int frx_traj_to_msg(int n_pieces, const double *T, const double *C, double *coef_x, double *coef_y, double *coef_z, double *time,
                    unsigned *order) {
    if (n_pieces <= 0 || !T || !C || !coef_x || !coef_y || !coef_z || !time || !order) return frx::set_error(FRX_ERR_INVALID_ARG, "frx_traj_to_msg: null or out-of-range argument");
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
    if (n_segment <= 0 || !coef_x || !coef_y || !coef_z || !time || !order || !pos || !vel || !acc || !jerk) return frx::set_error(FRX_ERR_INVALID_ARG, "frx_msg_sample: null or out-of-range argument");
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
