// Safe-flight-corridor cells on the device: LineSegment3D::dilate for a BATCH of path segments against one
// obstacle point cloud, one workgroup per segment.
//
// Reference (decomp_util, /root/reference/src/sfc/DecompROS/decomp_ros_utils/include/...):
//   decomp_base.h:35-40     set_obs: only the points inside the segment's local bounding box count
//   line_segment.h:47-85    add_local_bbox: six planes around the segment
//   line_segment.h:136-214  find_ellipsoid: shrink an ellipsoid around the segment until no obstacle point is inside it - every step is
//                           "closest point in the ellipsoid's metric" + "which points are still inside"
//   decomp_base.h:63-83     find_polyhedron: repeatedly take the closest remaining point, add the ellipsoid's tangent plane there,
//                           drop everything behind it
// Every step of those loops is a scan over the points: an arg-min and a filter.  Here the candidate points of a segment are compacted
// IN CLOUD ORDER into LDS once (the reference's tie rule is "first of the minima" in that order), the scans run on 256 lanes, and the
// few scalars of a step (3x3 shape matrices, trigonometry) are computed by one lane and broadcast - so a Monte-Carlo sweep can produce
// its corridors next to the optimiser's data instead of on the host (frx_geometry.cpp keeps the host form, same arithmetic).
#pragma once
#include <hip/hip_runtime.h>

#include "frx_wave.hpp"

namespace frx {

struct DilateArgs {
    const double *p1, *p2;       // [S][3] segment end points
    const double *obs;           // [n_obs][3]
    double bbox[3], offset;
    int S, n_obs, cap_planes, pcap;
    int *n_planes;               // [S]  (-1: more than pcap candidate points, -2: more than cap_planes planes)
    double *h_rec;               // [S][cap_planes][6]  (outer normal, point)
    double *ell_C, *ell_d;       // [S][9], [S][3] or null
};

namespace cg {
struct V3 { double x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 scl(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double nrm(V3 a) { return sqrt(dot(a, a)); }
struct M3 { double m[9]; };      // row-major
__device__ __forceinline__ V3 mul(const M3 &A, V3 v) { return {A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z, A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z}; }
__device__ __forceinline__ V3 mulT(const M3 &A, V3 v) { return {A.m[0] * v.x + A.m[3] * v.y + A.m[6] * v.z, A.m[1] * v.x + A.m[4] * v.y + A.m[7] * v.z, A.m[2] * v.x + A.m[5] * v.y + A.m[8] * v.z}; }
__device__ __forceinline__ M3 mul(const M3 &A, const M3 &B) {
    M3 C;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
__device__ __forceinline__ M3 transpose(const M3 &A) { M3 T; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T.m[3 * i + j] = A.m[3 * j + i]; return T; }
__device__ __forceinline__ M3 inverse(const M3 &A) {
    auto cof = [&](int i, int j) { const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return A.m[3 * i1 + j1] * A.m[3 * i2 + j2] - A.m[3 * i1 + j2] * A.m[3 * i2 + j1]; };
    const double det = cof(0, 0) * A.m[0] + cof(1, 0) * A.m[3] + cof(2, 0) * A.m[6];
    M3 I;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I.m[3 * j + i] = cof(i, j) / det;
    return I;
}
// rotation that takes e_x to the direction of v with zero roll (geometric_utils.h:27-35)
__device__ __forceinline__ M3 rotation_from_direction(V3 v) {
    const double pitch = atan2(-v.z, hypot(v.x, v.y)), yaw = atan2(v.y, v.x);
    const double cp = cos(pitch), sp = sin(pitch), cy = cos(yaw), sy = sin(yaw);
    return M3{{cy * cp, -sy, cy * sp, sy * cp, cy, sy * sp, -sp, 0.0, cp}};
}
__device__ __forceinline__ M3 roll_about_x(double roll) { const double c = cos(roll), s = sin(roll); return M3{{1.0, 0.0, 0.0, 0.0, c, -s, 0.0, s, c}}; }
// C = R diag(a) R^T and its inverse (ellipsoid.h)
__device__ __forceinline__ void set_shape(const M3 &R, double a0, double a1, double a2, M3 &C, M3 &Cinv) {
    const M3 D{{a0, 0.0, 0.0, 0.0, a1, 0.0, 0.0, 0.0, a2}};
    C = mul(mul(R, D), transpose(R));
    Cinv = inverse(C);
}
} // namespace cg

constexpr double kDecompEpsDev = 1e-10;                         // decomp_basis/data_type.h:129

// LDS (doubles): pts[pcap][3] | shape[32] (Cinv 0-8, d 9-11, scalar scratch) | box[6][6] | red[2 * 8] | then int: live[pcap], in0[pcap], cnt[257]
__global__ __launch_bounds__(256) void k_dilate(DilateArgs a) {
    using namespace cg;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int seg = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double *pts = sm, *shape = pts + (size_t)3 * a.pcap, *box = shape + 32, *red = box + 36;
    int *live = (int *)(red + 16), *in0 = live + a.pcap, *cnt = in0 + a.pcap;
    const V3 p1{a.p1[3 * seg], a.p1[3 * seg + 1], a.p1[3 * seg + 2]}, p2{a.p2[3 * seg], a.p2[3 * seg + 1], a.p2[3 * seg + 2]};
    // ---- local bounding box (line_segment.h:47-85) ----
    const bool has_box = sqrt(a.bbox[0] * a.bbox[0] + a.bbox[1] * a.bbox[1] + a.bbox[2] * a.bbox[2]) != 0.0;
    if (t == 0 && has_box) {
        V3 dir = sub(p2, p1); dir = scl(dir, 1.0 / nrm(dir));
        V3 h{dir.y, -dir.x, 0.0};
        if (nrm(h) == 0.0) h = V3{-1.0, 0.0, 0.0};
        h = scl(h, 1.0 / nrm(h));
        const V3 v{dir.y * h.z - dir.z * h.y, dir.z * h.x - dir.x * h.z, dir.x * h.y - dir.y * h.x};
        const V3 nn[6] = {h, scl(h, -1.0), dir, scl(dir, -1.0), v, scl(v, -1.0)};
        const V3 pp[6] = {add(p1, scl(h, a.bbox[1])), sub(p1, scl(h, a.bbox[1])), add(p2, scl(dir, a.bbox[0])), sub(p1, scl(dir, a.bbox[0])), add(p1, scl(v, a.bbox[2])), sub(p1, scl(v, a.bbox[2]))};
        for (int k = 0; k < 6; k++) { box[6 * k] = nn[k].x; box[6 * k + 1] = nn[k].y; box[6 * k + 2] = nn[k].z; box[6 * k + 3] = pp[k].x; box[6 * k + 4] = pp[k].y; box[6 * k + 5] = pp[k].z; }
    }
    __syncthreads();
    auto inside_box = [&](V3 q) {
        if (!has_box) return true;
        bool in = true;
#pragma unroll
        for (int k = 0; k < 6; k++) in = in && !(box[6 * k] * (q.x - box[6 * k + 3]) + box[6 * k + 1] * (q.y - box[6 * k + 4]) + box[6 * k + 2] * (q.z - box[6 * k + 5]) > kDecompEpsDev);
        return in;
    };
    // ---- candidate points, compacted in cloud order (thread t owns the contiguous chunk [t c, (t+1) c)) ----
    const int chunk = (a.n_obs + 255) / 256, i0 = t * chunk, i1 = min(i0 + chunk, a.n_obs);
    int mine = 0;
    for (int i = i0; i < i1; i++) mine += inside_box(V3{a.obs[3 * i], a.obs[3 * i + 1], a.obs[3 * i + 2]}) ? 1 : 0;
    cnt[t] = mine;
    __syncthreads();
    if (t == 0) { int s = 0; for (int k = 0; k < 256; k++) { const int c = cnt[k]; cnt[k] = s; s += c; } cnt[256] = s; }
    __syncthreads();
    const int M = cnt[256];
    if (M > a.pcap) { if (t == 0) a.n_planes[seg] = -1; return; }
    {
        int w = cnt[t];
        for (int i = i0; i < i1; i++) {
            const V3 q{a.obs[3 * i], a.obs[3 * i + 1], a.obs[3 * i + 2]};
            if (inside_box(q)) { pts[3 * w] = q.x; pts[3 * w + 1] = q.y; pts[3 * w + 2] = q.z; w++; }
        }
    }
    __syncthreads();
    // ---- helpers: distance in the ellipsoid's metric, block-wide arg-min over the live points (first of the minima) ----
    auto dist = [&](int i) {
        const V3 r{pts[3 * i] - shape[9], pts[3 * i + 1] - shape[10], pts[3 * i + 2] - shape[11]};
        const V3 u{shape[0] * r.x + shape[1] * r.y + shape[2] * r.z, shape[3] * r.x + shape[4] * r.y + shape[5] * r.z, shape[6] * r.x + shape[7] * r.y + shape[8] * r.z};
        return sqrt(u.x * u.x + u.y * u.y + u.z * u.z);
    };
    auto argmin_live = [&]() -> int {                              // returns -1 when nothing is alive; uniform over the block
        double best = 1.7976931348623157e308; int bi = 0x7fffffff;
        for (int i = t; i < M; i += 256) if (live[i]) { const double d = dist(i); if (d < best || (d == best && i < bi)) { best = d; bi = i; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { red[2 * wave] = best; red[2 * wave + 1] = (double)bi; }
        __syncthreads();
        double b0 = red[0]; int bb = (int)red[1];
#pragma unroll
        for (int w2 = 1; w2 < 4; w2++) { const double ob = red[2 * w2]; const int oi = (int)red[2 * w2 + 1]; if (ob < b0 || (ob == b0 && oi < bb)) { b0 = ob; bb = oi; } }
        __syncthreads();
        return bb == 0x7fffffff ? -1 : bb;
    };
    auto publish_shape = [&](const M3 &Cinv) { for (int k = 0; k < 9; k++) shape[k] = Cinv.m[k]; };
    // ---- find_ellipsoid (line_segment.h:136-214) ----
    const double f = nrm(sub(p1, p2)) / 2;
    double a0 = f + a.offset, a1 = f, a2 = f;
    if (a0 > 0) { const double ratio = a1 / a0; a0 *= ratio; a1 *= ratio; a2 *= ratio; }
    const M3 Ri = rotation_from_direction(sub(p2, p1));
    M3 Rf = Ri, C, Cinv;
    const V3 d = scl(add(p1, p2), 0.5);
    set_shape(Ri, a0, a1, a2, C, Cinv);
    if (t == 0) { publish_shape(Cinv); shape[9] = d.x; shape[10] = d.y; shape[11] = d.z; }
    __syncthreads();
    for (int i = t; i < M; i += 256) { const int in = dist(i) <= 1 ? 1 : 0; in0[i] = in; live[i] = in; }
    __syncthreads();
    for (;;) {                                                     // shrink the two short axes together, rolled towards the closest point
        const int ic = argmin_live();
        if (ic < 0) break;
        const V3 pw{pts[3 * ic], pts[3 * ic + 1], pts[3 * ic + 2]};
        V3 p = mulT(Ri, sub(pw, d));
        Rf = mul(Ri, roll_about_x(atan2(p.z, p.y)));
        p = mulT(Rf, sub(pw, d));
        if (p.x < a0) a1 = fabs(p.y) / sqrt(1 - (p.x / a0) * (p.x / a0));
        set_shape(Rf, a0, a1, a1, C, Cinv);
        if (t == 0) publish_shape(Cinv);
        __syncthreads();
        for (int i = t; i < M; i += 256) if (live[i] && !(1 - dist(i) > kDecompEpsDev)) live[i] = 0;
        __syncthreads();
    }
    set_shape(Rf, a0, a1, a2, C, Cinv);                            // the third axis starts again from its old length
    if (t == 0) publish_shape(Cinv);
    __syncthreads();
    for (int i = t; i < M; i += 256) live[i] = (in0[i] && dist(i) <= 1) ? 1 : 0;
    __syncthreads();
    for (;;) {
        const int ic = argmin_live();
        if (ic < 0) break;
        const V3 pw{pts[3 * ic], pts[3 * ic + 1], pts[3 * ic + 2]};
        const V3 p = mulT(Rf, sub(pw, d));
        const double dd = 1 - (p.x / a0) * (p.x / a0) - (p.y / a1) * (p.y / a1);
        if (dd > kDecompEpsDev) a2 = fabs(p.z) / sqrt(dd);
        set_shape(Rf, a0, a1, a2, C, Cinv);
        if (t == 0) publish_shape(Cinv);
        __syncthreads();
        for (int i = t; i < M; i += 256) if (live[i] && !(1 - dist(i) > kDecompEpsDev)) live[i] = 0;
        __syncthreads();
    }
    // ---- find_polyhedron (decomp_base.h:63-83) ----
    const M3 Q = mul(Cinv, transpose(Cinv));
    for (int i = t; i < M; i += 256) live[i] = 1;
    __syncthreads();
    int np = 0;
    double *out = a.h_rec + (size_t)seg * a.cap_planes * 6;
    for (;;) {
        const int ic = argmin_live();
        if (ic < 0) break;
        const V3 c{pts[3 * ic], pts[3 * ic + 1], pts[3 * ic + 2]};
        V3 n = mul(Q, sub(c, d)); n = scl(n, 1.0 / nrm(n));        // ellipsoid.h:53-58
        if (np >= a.cap_planes) { if (t == 0) a.n_planes[seg] = -2; return; }
        if (t == 0) { out[6 * np] = n.x; out[6 * np + 1] = n.y; out[6 * np + 2] = n.z; out[6 * np + 3] = c.x; out[6 * np + 4] = c.y; out[6 * np + 5] = c.z; }
        np++;
        for (int i = t; i < M; i += 256) if (live[i] && !(n.x * (pts[3 * i] - c.x) + n.y * (pts[3 * i + 1] - c.y) + n.z * (pts[3 * i + 2] - c.z) < 0)) live[i] = 0;
        __syncthreads();
    }
    if (has_box) {
        if (np + 6 > a.cap_planes) { if (t == 0) a.n_planes[seg] = -2; return; }
        if (t < 36) out[6 * np + t] = box[t];
        np += 6;
    }
    if (t == 0) {
        a.n_planes[seg] = np;
        if (a.ell_C) for (int k = 0; k < 9; k++) a.ell_C[9 * seg + k] = C.m[k];
        if (a.ell_d) { a.ell_d[3 * seg] = d.x; a.ell_d[3 * seg + 1] = d.y; a.ell_d[3 * seg + 2] = d.z; }
    }
}

} // namespace frx
