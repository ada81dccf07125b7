// TEST INFRASTRUCTURE.  Compiles the reference's safe-flight-corridor decomposition (decomp_util: line_segment.h,
// decomp_base.h, ellipsoid.h, polyhedron.h — header-only) UNMODIFIED from where it lies under /root/reference, against
// oracle/eigen_shim, into oracle/_ref/libref_decomp.so.  Used by tests/ and tests/golden/make_golden.py to pin the
// product's corridor generation (frx_line_segment_dilate / frx_corridor_generate); never by the product.
#include <decomp_util/ellipsoid_decomp.h>

extern "C" {

// LineSegment3D::dilate(offset) on segment p1-p2 with the obstacle cloud obs[3 n_obs] and local bounding box bbox
// (line_segment.h:31-35): hyperplanes (n, p) in the order the reference produces them, the ellipsoid (C row-major, d).
int ref_line_segment_dilate(const double *p1, const double *p2, const double *bbox, int n_obs, const double *obs, double offset, int cap,
                            int *n_planes, double *h_rec, double *ell_C, double *ell_d) {
    vec_Vec3f O;
    for (int i = 0; i < n_obs; i++) O.push_back(Vec3f(obs[3 * i], obs[3 * i + 1], obs[3 * i + 2]));
    LineSegment3D ls(Vec3f(p1[0], p1[1], p1[2]), Vec3f(p2[0], p2[1], p2[2]));
    ls.set_local_bbox(Vec3f(bbox[0], bbox[1], bbox[2]));
    ls.set_obs(O);
    ls.dilate(offset);
    const auto hs = ls.get_polyhedron().hyperplanes();
    *n_planes = (int)hs.size();
    if ((int)hs.size() > cap) return -1;
    for (size_t k = 0; k < hs.size(); k++)
        for (int a = 0; a < 3; a++) { h_rec[6 * k + a] = hs[k].n_(a); h_rec[6 * k + 3 + a] = hs[k].p_(a); }
    const auto E = ls.get_ellipsoid();
    for (int i = 0; i < 3; i++) { ell_d[i] = E.d_(i); for (int j = 0; j < 3; j++) ell_C[3 * i + j] = E.C_(i, j); }
    return 0;
}

// EllipsoidDecomp3D::dilate on a two-point path, as MavGlobalPlanner::plan calls it (MinCoPlan_CPU.cpp:58-62)
int ref_decomp_dilate(const double *p1, const double *p2, const double *bbox, int n_obs, const double *obs, int cap, int *n_planes, double *h_rec) {
    vec_Vec3f O, line;
    for (int i = 0; i < n_obs; i++) O.push_back(Vec3f(obs[3 * i], obs[3 * i + 1], obs[3 * i + 2]));
    line.push_back(Vec3f(p1[0], p1[1], p1[2])); line.push_back(Vec3f(p2[0], p2[1], p2[2]));
    EllipsoidDecomp3D dec;
    dec.set_obs(O);
    dec.set_local_bbox(Vec3f(bbox[0], bbox[1], bbox[2]));
    dec.dilate(line);
    const auto hs = dec.get_polyhedrons()[0].hyperplanes();
    *n_planes = (int)hs.size();
    if ((int)hs.size() > cap) return -1;
    for (size_t k = 0; k < hs.size(); k++)
        for (int a = 0; a < 3; a++) { h_rec[6 * k + a] = hs[k].n_(a); h_rec[6 * k + 3 + a] = hs[k].p_(a); }
    return 0;
}

// Polyhedron3D::inside (polyhedron.h: signed distance > 1e-10 rejects)
int ref_poly_inside(int n_planes, const double *h_rec, const double *pt) {
    Polyhedron3D poly;
    for (int k = 0; k < n_planes; k++) poly.add(Hyperplane3D(Vec3f(h_rec[6 * k + 3], h_rec[6 * k + 4], h_rec[6 * k + 5]), Vec3f(h_rec[6 * k], h_rec[6 * k + 1], h_rec[6 * k + 2])));
    return poly.inside(Vec3f(pt[0], pt[1], pt[2])) ? 1 : 0;
}
}
