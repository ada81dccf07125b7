// TEST INFRASTRUCTURE.  Compiles the reference's JPSPlanner<3> and MapUtil<3> where they lie
// (/root/reference/src/path_searching/{src/jps_planner/jps_planner.cpp, src/jps_planner/graph_search.cpp, include/jps_collision/map_util.h, ...})
// (graph_search.cpp is a translation unit of its own, as in the reference's build: jps_planner.h declares a global ::GraphSearch next to JPS::GraphSearch)
// against oracle/eigen_shim, oracle/boost_shim and oracle/ros_shim (Eigen, Boost, ROS, PCL and octomap are absent from the image) and
// exposes them through a C ABI for tests/test_front_end.py.  Nothing of the reference is copied into the repository.
#include <jps_planner/jps_planner/jps_planner.h>
#include "../src/jps_planner/jps_planner.cpp"

#include <cstring>
#include <memory>

namespace {
struct Ctx {
    std::shared_ptr<JPS::VoxelMapUtil> map = std::make_shared<JPS::VoxelMapUtil>();
    JPSPlanner3D planner{false};
};
int put(const vec_Vecf<3> &p, double *out, int cap) {
    for (int i = 0; i < int(p.size()) && i < cap; i++) { out[3 * i] = p[i](0); out[3 * i + 1] = p[i](1); out[3 * i + 2] = p[i](2); }
    return int(p.size());
}
} // namespace

extern "C" {
void *ref_jp_create(const double *origin, const int *dim, const signed char *cells, double res) {
    Ctx *c = new Ctx;
    JPS::Tmap m(cells, cells + size_t(dim[0]) * dim[1] * dim[2]);
    c->map->setMap(Vec3f(origin[0], origin[1], origin[2]), Vec3i(dim[0], dim[1], dim[2]), m, res);      // MapUtil::setMap, map_util.h:364
    c->planner.setMapUtil(c->map);
    c->planner.updateMap();
    return c;
}
void ref_jp_destroy(void *h) { delete (Ctx *)h; }
// MapUtil::setObs (map_util.h:109): marks the cell of a cloud point; returns the map afterwards
void ref_jp_mark(void *h, const double *pts, int n, signed char *cells_out) {
    Ctx *c = (Ctx *)h;
    for (int i = 0; i < n; i++) c->map->setObs(Eigen::Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    const JPS::Tmap m = c->map->getMap();
    std::memcpy(cells_out, m.data(), m.size());
    c->planner.updateMap();
}
int ref_jp_is_blocked(void *h, const double *p1, const double *p2) {
    Ctx *c = (Ctx *)h;
    return c->map->isBlocked(Vec3f(p1[0], p1[1], p1[2]), Vec3f(p2[0], p2[1], p2[2])) ? 1 : 0;
}
void ref_jp_float_to_int(void *h, const double *p, int *out) {
    const Vec3i v = ((Ctx *)h)->map->floatToInt(Vec3f(p[0], p[1], p[2]));
    out[0] = v(0); out[1] = v(1); out[2] = v(2);
}
// JPSPlanner<3>::plan (jps_planner.cpp:333): returns its bool, *status = status(); the three paths as the planner holds them afterwards
int ref_jp_plan(void *h, const double *start, const double *goal, double eps, int use_jps, int *status, double *raw, int *n_raw, double *path, int *n_path,
                double *sample, int *n_sample, int cap) {
    Ctx *c = (Ctx *)h;
    const bool ok = c->planner.plan(Vec3f(start[0], start[1], start[2]), Vec3f(goal[0], goal[1], goal[2]), eps, use_jps != 0);
    *status = c->planner.status();
    *n_raw = put(c->planner.getRawPath(), raw, cap);
    *n_path = put(c->planner.getPath(), path, cap);
    *n_sample = ok ? put(c->planner.getSamplePath(), sample, cap) : 0;
    return ok ? 1 : 0;
}
}
