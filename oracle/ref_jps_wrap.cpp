// TEST INFRASTRUCTURE.  Compiles the reference's own grid search where it lies
// (/root/reference/src/path_searching/src/jps_planner/graph_search.cpp + include/jps_planner/jps_planner/graph_search.h)
// against oracle/boost_shim (Boost is absent from the image; see boost_shim/boost/heap/d_ary_heap.hpp) and exposes it through
// a C ABI for tests/test_front_end.py.  Nothing of the reference is copied into the repository: the source is #included from
// the reference tree at build time and only the binary lands in oracle/_ref/ (git-ignored).
#include <jps_planner/jps_planner/graph_search.h>
#include "../src/jps_planner/graph_search.cpp"

#include <cstring>

extern "C" {

// GraphSearch::plan (graph_search.cpp:79,104).  dim[2] == 0 selects the 2-D constructor.  The path comes back as the
// reference holds it (goal first, graph_search.cpp:239-248); returns 1 when a path was found, 0 when not, and the number of
// states in the closed set in *n_closed (getCloseSet, :497).
int ref_grid_search(const char *cmap, const int *dim, const int *start, const int *goal, double eps, int use_jps,
                    int max_expand, int *path_xyz, int cap, int *n_path, int *n_closed, double *goal_g) {
  bool ok;
  JPS::GraphSearch *gs;
  if (dim[2] == 0) {
    gs = new JPS::GraphSearch(cmap, dim[0], dim[1], eps, false);
    ok = gs->plan(start[0], start[1], goal[0], goal[1], use_jps != 0, max_expand);
  } else {
    gs = new JPS::GraphSearch(cmap, dim[0], dim[1], dim[2], eps, false);
    ok = gs->plan(start[0], start[1], start[2], goal[0], goal[1], goal[2], use_jps != 0, max_expand);
  }
  auto path = gs->getPath();
  *n_path = int(path.size());
  for (int i = 0; i < int(path.size()) && i < cap; i++) {
    path_xyz[3 * i] = path[i]->x;
    path_xyz[3 * i + 1] = path[i]->y;
    path_xyz[3 * i + 2] = path[i]->z;
  }
  if (goal_g) *goal_g = path.empty() ? -1.0 : path.front()->g;
  if (n_closed) *n_closed = int(gs->getCloseSet().size());
  delete gs;
  return ok ? 1 : 0;
}

// The reference's jump-point neighbour tables (graph_search.h:72-127), flattened as they are declared.
void ref_jps_tables(int *ns3, int *f13, int *f23, int *ns2, int *f12, int *f22) {
  JPS::JPS3DNeib t3;
  std::memset(ns3, 0, sizeof(t3.ns));
  JPS::JPS2DNeib t2;
  std::memcpy(ns3, t3.ns, sizeof(t3.ns));
  std::memcpy(f13, t3.f1, sizeof(t3.f1));
  std::memcpy(f23, t3.f2, sizeof(t3.f2));
  std::memcpy(ns2, t2.ns, sizeof(t2.ns));
  std::memcpy(f12, t2.f1, sizeof(t2.f1));
  std::memcpy(f22, t2.f2, sizeof(t2.f2));
}
}
