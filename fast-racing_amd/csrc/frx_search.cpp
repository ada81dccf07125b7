// Path-search front end on the voxel grid: the step of MavGlobalPlanner::plan upstream of the corridor
// (MinCoPlan_CPU.cpp:13-35).  Host code by design: a best-first search is one long dependent chain of heap operations, there
// is nothing in it for 256 CUs, and the reference's own tie order (which decides WHICH of the equally short grid paths comes
// back, and with it every corridor cell downstream) is defined by the exact sequence of those heap operations.
//
//   GridSearch            = JPS::GraphSearch (graph_search.h:129-264, graph_search.cpp:6-236): A* on the dense occupancy array
//                           with the reference's successor order, its tolerance comparator (graph_search.h:20-33) and an
//                           indexed binary heap that performs the sift steps of the container the reference uses
//                           (boost d_ary_heap, arity 2, mutable: push = append + sift up; pop = last to the root + sift down;
//                           increase = sift up).
//   jump-point tables     = JPS3DNeib / JPS2DNeib (graph_search.h:72-127).  The reference spells them as switch statements
//                           (graph_search.cpp:541-946); here they are generated from the rules those statements encode
//                           (natural neighbours = non-empty sub-directions of the move; forced neighbours = a blocked cell
//                           beside the move opens the cell behind it) and tests/test_front_end.py compares all 27x(26+12+12)
//                           entries with the compiled reference.
//   VoxelMap              = the queries of JPS::MapUtil<3> the planner needs (map_util.h:320-425).
//   plan_leg / route      = JPSPlanner<3>::plan (jps_planner.cpp:333-420) with removeCornerPts (:54-95), removeLinePts (:98-117)
//                           and samplePath (:118-148); the gate-to-gate concatenation of MinCoPlan_CPU.cpp:19-35.
//
// This file is compiled with -ffp-contract=off: cell indices come from std::round((p - origin) / res - 0.5) and the sample
// points of samplePath fall exactly on cell faces (0.05 m steps on a 0.1 m grid), so a fused multiply-add would move points
// across faces.  The test suite's restatement evaluates the same expressions in IEEE double without fusing.
#include "../../include/frx.h"
#include "../../include/frx_debug.h"
#include "frx_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// jump-point neighbour tables
// ---------------------------------------------------------------------------------------------------------------------------
struct Neib3 {
    // index (dx+1) + 3 (dy+1) + 9 (dz+1); counts by the 1-norm of the move (graph_search.h:111: {26,0},{1,8},{3,12},{7,12})
    signed char ns[27][26][3];
    signed char f1[27][12][3];
    signed char f2[27][12][3];
    Neib3();
};
struct Neib2 {
    signed char ns[9][8][2];
    signed char f1[9][2][2];
    signed char f2[9][2][2];
    Neib2();
};
constexpr int NSZ3[4][2] = {{26, 0}, {1, 8}, {3, 12}, {7, 12}};
constexpr int NSZ2[3][2] = {{8, 0}, {1, 2}, {3, 2}};

// the order in which the reference walks an offset: 0, +1, -1
constexpr int ZPM[3] = {0, 1, -1};

inline void put(signed char *dst, int x, int y, int z) { dst[0] = (signed char)x; dst[1] = (signed char)y; dst[2] = (signed char)z; }

Neib3::Neib3() {
    std::memset(this, 0, sizeof(*this));
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int id = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
                const int d[3] = {dx, dy, dz};
                const int norm1 = std::abs(dx) + std::abs(dy) + std::abs(dz);
                if (norm1 == 0) {  // at the start every neighbour is natural: layers z = 0, +1, -1, rows y = 0, +1, -1
                    int k = 0;
                    for (int iz = 0; iz < 3; iz++)
                        for (int iy = 0; iy < 3; iy++)
                            for (int ix = 0; ix < 3; ix++) {
                                if (!ix && !iy && !iz) continue;
                                put(ns[id][k++], ZPM[ix], ZPM[iy], ZPM[iz]);
                            }
                } else if (norm1 == 1) {
                    put(ns[id][0], dx, dy, dz);
                    // the eight cells around the axis of the move, in the frame (u, v) = the two other axes; a blocked one
                    // forces the cell one step further along the move.  The reference lists (u, v) as below with
                    // u = x, v = y for a z move, u = z, v = y for an x move and u = x, v = z for a y move.
                    static const int uv[8][2] = {{0, 1}, {0, -1}, {1, 0}, {1, 1}, {1, -1}, {-1, 0}, {-1, 1}, {-1, -1}};
                    const int ua = dz ? 0 : (dx ? 2 : 0), va = dz ? 1 : (dx ? 1 : 2);
                    for (int k = 0; k < 8; k++) {
                        int f[3] = {0, 0, 0};
                        f[ua] = uv[k][0];
                        f[va] = uv[k][1];
                        put(f1[id][k], f[0], f[1], f[2]);
                        put(f2[id][k], f[0] + dx, f[1] + dy, f[2] + dz);
                    }
                } else if (norm1 == 2) {
                    // plane move: a, b = the axes it moves along, c = the axis it does not
                    int a, b, c;
                    if (dz == 0) a = 0, b = 1, c = 2;
                    else if (dx == 0) a = 1, b = 2, c = 0;
                    else a = 0, b = 2, c = 1;
                    int t[3] = {0, 0, 0};
                    t[b] = d[b];
                    put(ns[id][0], t[0], t[1], t[2]);
                    t[b] = 0, t[a] = d[a];
                    put(ns[id][1], t[0], t[1], t[2]);
                    put(ns[id][2], dx, dy, dz);
                    auto entry = [&](int k, int fa, int fb, int fc, int na, int nb, int nc) {
                        int f[3], n[3];
                        f[a] = fa, f[b] = fb, f[c] = fc;
                        n[a] = na, n[b] = nb, n[c] = nc;
                        put(f1[id][k], f[0], f[1], f[2]);
                        put(f2[id][k], n[0], n[1], n[2]);
                    };
                    const int A = d[a], B = d[b];
                    entry(0, 0, -B, 0, A, -B, 0);     // in-plane: behind along b
                    entry(1, -A, 0, 0, -A, B, 0);     // in-plane: behind along a
                    entry(2, 0, 0, 1, A, B, 1);       // above / below the plane
                    entry(3, 0, 0, -1, A, B, -1);
                    entry(4, 0, -B, 1, A, -B, 1);     // the in-plane pair, lifted and lowered
                    entry(5, -A, 0, 1, -A, B, 1);
                    entry(6, 0, -B, -1, A, -B, -1);
                    entry(7, -A, 0, -1, -A, B, -1);
                    entry(8, 0, 0, 1, A, 0, 1);       // extras opened by the cell above / below
                    entry(9, 0, 0, 1, 0, B, 1);
                    entry(10, 0, 0, -1, A, 0, -1);
                    entry(11, 0, 0, -1, 0, B, -1);
                } else {
                    // space diagonal: natural = axes, then planes, then the move itself
                    put(ns[id][0], dx, 0, 0);
                    put(ns[id][1], 0, dy, 0);
                    put(ns[id][2], 0, 0, dz);
                    put(ns[id][3], dx, dy, 0);
                    put(ns[id][4], dx, 0, dz);
                    put(ns[id][5], 0, dy, dz);
                    put(ns[id][6], dx, dy, dz);
                    auto entry = [&](int k, int fx, int fy, int fz, int nx, int ny, int nz) {
                        put(f1[id][k], fx, fy, fz);
                        put(f2[id][k], nx, ny, nz);
                    };
                    entry(0, -dx, 0, 0, -dx, dy, dz);   // behind along one axis
                    entry(1, 0, -dy, 0, dx, -dy, dz);
                    entry(2, 0, 0, -dz, dx, dy, -dz);
                    entry(3, 0, -dy, -dz, dx, -dy, -dz);  // behind along two (the last of the six that stop a jump)
                    entry(4, -dx, 0, -dz, -dx, dy, -dz);
                    entry(5, -dx, -dy, 0, -dx, -dy, dz);
                    entry(6, -dx, 0, 0, -dx, 0, dz);    // extras: each single-axis blocker also opens the two plane cells
                    entry(7, -dx, 0, 0, -dx, dy, 0);
                    entry(8, 0, -dy, 0, 0, -dy, dz);
                    entry(9, 0, -dy, 0, dx, -dy, 0);
                    entry(10, 0, 0, -dz, 0, dy, -dz);
                    entry(11, 0, 0, -dz, dx, 0, -dz);
                }
            }
}

Neib2::Neib2() {
    std::memset(this, 0, sizeof(*this));
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            const int id = (dx + 1) + 3 * (dy + 1);
            const int norm1 = std::abs(dx) + std::abs(dy);
            auto put2 = [](signed char *dst, int x, int y) { dst[0] = (signed char)x; dst[1] = (signed char)y; };
            if (norm1 == 0) {
                int k = 0;
                for (int iy = 0; iy < 3; iy++)
                    for (int ix = 0; ix < 3; ix++) {
                        if (!ix && !iy) continue;
                        put2(ns[id][k++], ZPM[ix], ZPM[iy]);
                    }
            } else if (norm1 == 1) {
                put2(ns[id][0], dx, dy);
                for (int k = 0; k < 2; k++) {
                    const int side = k == 0 ? 1 : -1;  // the two cells beside the move
                    const int fx = dx == 0 ? side : 0, fy = dx == 0 ? 0 : side;
                    put2(f1[id][k], fx, fy);
                    put2(f2[id][k], dx + fx, dy + fy);
                }
            } else {
                put2(ns[id][0], dx, 0);
                put2(ns[id][1], 0, dy);
                put2(ns[id][2], dx, dy);
                put2(f1[id][0], -dx, 0);
                put2(f2[id][0], -dx, dy);
                put2(f1[id][1], 0, -dy);
                put2(f2[id][1], dx, -dy);
            }
        }
}

const Neib3 &neib3() {
    static const Neib3 t;
    return t;
}
const Neib2 &neib2() {
    static const Neib2 t;
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------------
// the grid search
// ---------------------------------------------------------------------------------------------------------------------------
struct Node {
    double g, h;
    int id, parent;
    int x, y, z;
    int heap_pos;
    signed char dx, dy, dz;
    bool opened, closed;
};

class GridSearch {
   public:
    GridSearch(const signed char *cmap, int X, int Y, int Z, double eps)
        : cmap_(cmap), X_(X), Y_(Y), Z_(Z > 0 ? Z : 1), two_d_(Z <= 0), eps_(eps) {
        // one 32-bit slot per cell (0 = never seen), zero pages until touched: a 700 x 4000 x 28 map costs what the search visits
        slot_ = static_cast<uint32_t *>(std::calloc(size_t(X_) * Y_ * Z_, sizeof(uint32_t)));
    }
    ~GridSearch() { std::free(slot_); }
    GridSearch(const GridSearch &) = delete;
    GridSearch &operator=(const GridSearch &) = delete;
    bool allocated() const { return slot_ != nullptr; }

    // graph_search.cpp:79-236.  true when the goal was closed; path() then runs goal -> start.
    bool plan(const int *s, const int *g, bool use_jps, int max_expand);
    const std::vector<int> &path() const { return path_; }  // node indices
    const Node &node(int i) const { return nodes_[i]; }
    int expanded() const { return expanded_; }
    bool broken_chain() const { return broken_chain_; }

   private:
    int coord_id(int x, int y, int z) const { return x + y * X_ + z * X_ * Y_; }
    bool inside(int x, int y, int z) const { return x >= 0 && x < X_ && y >= 0 && y < Y_ && z >= 0 && z < Z_; }
    bool is_free(int x, int y, int z) const { return inside(x, y, z) && cmap_[coord_id(x, y, z)] == 0; }
    // outside the map is neither free nor occupied (graph_search.cpp:52-70): the border never forces a neighbour
    bool is_occupied(int x, int y, int z) const { return inside(x, y, z) && cmap_[coord_id(x, y, z)] > 0; }
    double heur(int x, int y, int z) const {
        const int ax = x - gx_, ay = y - gy_, az = z - gz_;
        return eps_ * std::sqrt(double(ax * ax + ay * ay + az * az));
    }

    // "a is served after b" (graph_search.h:20-33): larger f, or within 1e-6 of it the smaller g
    bool after(int a, int b) const {
        const Node &na = nodes_[a], &nb = nodes_[b];
        const double f1 = na.g + na.h, f2 = nb.g + nb.h;
        if (f1 >= f2 - 0.000001 && f1 <= f2 + 0.000001) return na.g < nb.g;
        return f1 > f2;
    }
    void heap_swap(size_t i, size_t j) {
        std::swap(heap_[i], heap_[j]);
        nodes_[heap_[i]].heap_pos = int(i);
        nodes_[heap_[j]].heap_pos = int(j);
    }
    void sift_up(size_t i) {
        while (i != 0) {
            const size_t p = (i - 1) / 2;
            if (!after(heap_[p], heap_[i])) return;
            heap_swap(p, i);
            i = p;
        }
    }
    void sift_down(size_t i) {
        for (;;) {
            const size_t l = 2 * i + 1;
            if (l >= heap_.size()) return;
            size_t c = l;  // the first child unless the second is strictly ahead of it
            if (l + 1 < heap_.size() && after(heap_[l], heap_[l + 1])) c = l + 1;
            if (after(heap_[c], heap_[i])) return;  // ties move down, as in the reference's container
            heap_swap(c, i);
            i = c;
        }
    }
    void heap_push(int n) {
        heap_.push_back(n);
        nodes_[n].heap_pos = int(heap_.size() - 1);
        sift_up(heap_.size() - 1);
    }
    int heap_pop() {
        const int top = heap_.front();
        heap_swap(0, heap_.size() - 1);
        heap_.pop_back();
        if (!heap_.empty()) sift_down(0);
        return top;
    }

    // the node of a cell, created on first sight with the move that found it (graph_search.cpp:262-266)
    int touch(int x, int y, int z, int dx, int dy, int dz) {
        const int id = coord_id(x, y, z);
        if (slot_[id]) return int(slot_[id] - 1);
        Node n;
        n.g = std::numeric_limits<double>::infinity();
        n.h = heur(x, y, z);
        n.id = id, n.parent = -1;
        n.x = x, n.y = y, n.z = z;
        n.heap_pos = -1;
        n.dx = (signed char)dx, n.dy = (signed char)dy, n.dz = (signed char)dz;
        n.opened = n.closed = false;
        nodes_.push_back(n);
        slot_[id] = uint32_t(nodes_.size());
        return int(nodes_.size() - 1);
    }

    void successors(int cur);
    void jps_successors(int cur);
    bool jump3(int x, int y, int z, int dx, int dy, int dz, int *o);
    bool jump2(int x, int y, int dx, int dy, int *o);
    bool forced3(int x, int y, int z, int dx, int dy, int dz) const;
    bool forced2(int x, int y, int dx, int dy) const;

    const signed char *cmap_;
    int X_, Y_, Z_;
    bool two_d_;
    double eps_;
    int gx_ = 0, gy_ = 0, gz_ = 0;
    uint32_t *slot_ = nullptr;
    std::vector<Node> nodes_;
    std::vector<int> heap_;
    std::vector<int> succ_;
    std::vector<double> succ_cost_;
    std::vector<int> path_;
    int expanded_ = 0;
    bool broken_chain_ = false;
};

void GridSearch::successors(int cur) {
    const int cx = nodes_[cur].x, cy = nodes_[cur].y, cz = nodes_[cur].z;
    if (two_d_) {
        // all eight neighbours, x outermost (graph_search.cpp:12-17)
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++) {
                if (!dx && !dy) continue;
                if (!is_free(cx + dx, cy + dy, 0)) continue;
                succ_.push_back(touch(cx + dx, cy + dy, 0, dx, dy, 0));
                succ_cost_.push_back(std::sqrt(double(dx * dx + dy * dy)));
            }
    } else {
        // in 3-D the reference keeps only the six face neighbours (graph_search.cpp:29-38 skips |d|_1 >= 2), x outermost
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dz = -1; dz <= 1; dz++) {
                    if (std::abs(dx) + std::abs(dy) + std::abs(dz) != 1) continue;
                    if (!is_free(cx + dx, cy + dy, cz + dz)) continue;
                    succ_.push_back(touch(cx + dx, cy + dy, cz + dz, dx, dy, dz));
                    succ_cost_.push_back(std::sqrt(double(dx * dx + dy * dy + dz * dz)));
                }
    }
}

bool GridSearch::forced3(int x, int y, int z, int dx, int dy, int dz) const {
    const int norm1 = std::abs(dx) + std::abs(dy) + std::abs(dz);
    const int id = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
    const int n = norm1 == 3 ? 6 : (norm1 == 0 ? 0 : 8);  // graph_search.cpp:439-482
    const Neib3 &t = neib3();
    for (int k = 0; k < n; k++)
        if (is_occupied(x + t.f1[id][k][0], y + t.f1[id][k][1], z + t.f1[id][k][2])) return true;
    return false;
}

bool GridSearch::forced2(int x, int y, int dx, int dy) const {
    const int id = (dx + 1) + 3 * (dy + 1);
    const Neib2 &t = neib2();
    for (int k = 0; k < 2; k++)  // graph_search.cpp:426-436
        if (is_occupied(x + t.f1[id][k][0], y + t.f1[id][k][1], 0)) return true;
    return false;
}

// graph_search.cpp:394-423.  The reference recurses once per cell along the move; the walk along the move is a loop here and
// only the probes of the sub-directions recurse (depth <= 2).
bool GridSearch::jump3(int x, int y, int z, int dx, int dy, int dz, int *o) {
    const int id = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
    const int norm1 = std::abs(dx) + std::abs(dy) + std::abs(dz);
    const int nn = NSZ3[norm1][0];
    const Neib3 &t = neib3();
    for (;;) {
        x += dx, y += dy, z += dz;
        if (!is_free(x, y, z)) return false;
        o[0] = x, o[1] = y, o[2] = z;
        if (x == gx_ && y == gy_ && z == gz_) return true;
        if (forced3(x, y, z, dx, dy, dz)) return true;
        for (int k = 0; k < nn - 1; k++) {
            int tmp[3];
            if (jump3(x, y, z, t.ns[id][k][0], t.ns[id][k][1], t.ns[id][k][2], tmp)) return true;
        }
    }
}

bool GridSearch::jump2(int x, int y, int dx, int dy, int *o) {  // graph_search.cpp:367-391
    const int id = (dx + 1) + 3 * (dy + 1);
    const int nn = NSZ2[std::abs(dx) + std::abs(dy)][0];
    const Neib2 &t = neib2();
    for (;;) {
        x += dx, y += dy;
        if (!is_free(x, y, 0)) return false;
        o[0] = x, o[1] = y, o[2] = 0;
        if (x == gx_ && y == gy_) return true;
        if (forced2(x, y, dx, dy)) return true;
        for (int k = 0; k < nn - 1; k++) {
            int tmp[3];
            if (jump2(x, y, t.ns[id][k][0], t.ns[id][k][1], tmp)) return true;
        }
    }
}

void GridSearch::jps_successors(int cur) {  // graph_search.cpp:279-364
    const Node c = nodes_[cur];
    if (two_d_) {
        const int norm1 = std::abs(c.dx) + std::abs(c.dy);
        const int nn = NSZ2[norm1][0], nf = NSZ2[norm1][1];
        const int id = (c.dx + 1) + 3 * (c.dy + 1);
        const Neib2 &t = neib2();
        for (int dev = 0; dev < nn + nf; dev++) {
            int dx, dy, o[3];
            if (dev < nn) {
                dx = t.ns[id][dev][0], dy = t.ns[id][dev][1];
            } else {
                const int k = dev - nn;
                if (!is_occupied(c.x + t.f1[id][k][0], c.y + t.f1[id][k][1], 0)) continue;
                dx = t.f2[id][k][0], dy = t.f2[id][k][1];
            }
            if (!jump2(c.x, c.y, dx, dy, o)) continue;
            succ_.push_back(touch(o[0], o[1], 0, dx, dy, 0));
            const int ax = o[0] - c.x, ay = o[1] - c.y;
            succ_cost_.push_back(std::sqrt(double(ax * ax + ay * ay)));
        }
    } else {
        const int norm1 = std::abs(c.dx) + std::abs(c.dy) + std::abs(c.dz);
        const int nn = NSZ3[norm1][0], nf = NSZ3[norm1][1];
        const int id = (c.dx + 1) + 3 * (c.dy + 1) + 9 * (c.dz + 1);
        const Neib3 &t = neib3();
        for (int dev = 0; dev < nn + nf; dev++) {
            int dx, dy, dz, o[3];
            if (dev < nn) {
                dx = t.ns[id][dev][0], dy = t.ns[id][dev][1], dz = t.ns[id][dev][2];
            } else {
                const int k = dev - nn;
                if (!is_occupied(c.x + t.f1[id][k][0], c.y + t.f1[id][k][1], c.z + t.f1[id][k][2])) continue;
                dx = t.f2[id][k][0], dy = t.f2[id][k][1], dz = t.f2[id][k][2];
            }
            if (!jump3(c.x, c.y, c.z, dx, dy, dz, o)) continue;
            succ_.push_back(touch(o[0], o[1], o[2], dx, dy, dz));
            const int ax = o[0] - c.x, ay = o[1] - c.y, az = o[2] - c.z;
            succ_cost_.push_back(std::sqrt(double(ax * ax + ay * ay + az * az)));
        }
    }
}

bool GridSearch::plan(const int *s, const int *g, bool use_jps, int max_expand) {
    gx_ = g[0], gy_ = g[1], gz_ = two_d_ ? 0 : g[2];
    const int sz = two_d_ ? 0 : s[2];
    const int goal_id = coord_id(gx_, gy_, gz_);
    const int start_id = coord_id(s[0], s[1], sz);
    path_.clear();
    int cur = touch(s[0], s[1], sz, 0, 0, 0);
    nodes_[cur].g = 0.0;
    heap_push(cur);
    nodes_[cur].opened = true;

    expanded_ = 0;
    for (;;) {
        expanded_++;
        cur = heap_pop();
        nodes_[cur].closed = true;
        if (nodes_[cur].id == goal_id) break;

        succ_.clear();
        succ_cost_.clear();
        if (use_jps) jps_successors(cur);
        else successors(cur);

        for (size_t k = 0; k < succ_.size(); k++) {
            Node &ch = nodes_[succ_[k]];
            const Node &cn = nodes_[cur];
            const double tentative = cn.g + succ_cost_[k];
            if (tentative < ch.g) {
                // the reference rewrites parent and g BEFORE it looks at the lists (graph_search.cpp:170-173), so with an
                // inflated heuristic a closed cell can be re-parented without being re-expanded; the chain walk below follows
                // whatever that leaves.
                ch.parent = cn.id;
                ch.g = tentative;
                if (ch.opened && !ch.closed) {
                    sift_up(size_t(ch.heap_pos));
                    auto sgn = [](int v) { return (v > 0) - (v < 0); };
                    ch.dx = (signed char)sgn(ch.x - cn.x);
                    ch.dy = (signed char)sgn(ch.y - cn.y);
                    ch.dz = (signed char)sgn(ch.z - cn.z);
                } else if (ch.opened && ch.closed) {
                    continue;
                } else {
                    heap_push(succ_[k]);
                    nodes_[succ_[k]].opened = true;
                }
            }
        }
        if (max_expand > 0 && expanded_ >= max_expand) return false;
        if (heap_.empty()) return false;
    }

    // recoverPath (graph_search.cpp:239-248): goal first.  A parent chain can only fail to reach the start after the
    // re-parenting above; the reference would walk it for ever, this reports it.
    int n = cur;
    path_.push_back(n);
    size_t guard = nodes_.size() + 1;
    while (nodes_[n].id != start_id) {
        if (nodes_[n].parent < 0 || guard-- == 0) {
            broken_chain_ = true;
            path_.clear();
            return false;
        }
        n = int(slot_[nodes_[n].parent] - 1);
        path_.push_back(n);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// map queries and the planner on top of the search
// ---------------------------------------------------------------------------------------------------------------------------
struct V3 {
    double x, y, z;
};
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double norm(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

struct VoxelMap {
    const frx_voxel_map *m;
    bool outside(const int *p) const {
        for (int i = 0; i < 3; i++)
            if (p[i] < 0 || p[i] >= m->dim[i]) return true;
        return false;
    }
    size_t index(const int *p) const { return size_t(p[0]) + size_t(m->dim[0]) * p[1] + size_t(m->dim[0]) * m->dim[1] * p[2]; }
    bool is_free(const int *p) const { return !outside(p) && m->cells[index(p)] == 0; }  // map_util.h:336-341
    void to_cell(V3 p, int *c) const {                                                    // floatToInt, map_util.h:382-387
        c[0] = int(std::round((p.x - m->origin[0]) / m->res - 0.5));
        c[1] = int(std::round((p.y - m->origin[1]) / m->res - 0.5));
        c[2] = int(std::round((p.z - m->origin[2]) / m->res - 0.5));
    }
    V3 to_point(const int *c) const {  // intToFloat, map_util.h:389-392: the cell centre
        return {(double(c[0]) + 0.5) * m->res + m->origin[0], (double(c[1]) + 0.5) * m->res + m->origin[1],
                (double(c[2]) + 0.5) * m->res + m->origin[2]};
    }
    // rayTrace + isBlocked (map_util.h:395-425): cells at 0.8-cell steps strictly between the end points, stop at the border
    bool blocked(V3 a, V3 b, int val = 100) const {
        const V3 diff = b - a;
        const double dmax = std::max(std::fabs(diff.x / m->res), std::max(std::fabs(diff.y / m->res), std::fabs(diff.z / m->res)));
        const int max_diff = int(dmax / 0.8);
        const double s = 1.0 / max_diff;
        const V3 step = {diff.x * s, diff.y * s, diff.z * s};
        int prev[3] = {-1, -1, -1};
        for (int n = 1; n < max_diff; n++) {
            const V3 pt = {a.x + step.x * n, a.y + step.y * n, a.z + step.z * n};
            int c[3];
            to_cell(pt, c);
            if (outside(c)) break;
            if (c[0] != prev[0] || c[1] != prev[1] || c[2] != prev[2])
                if (m->cells[index(c)] >= val) return true;
            prev[0] = c[0], prev[1] = c[1], prev[2] = c[2];
        }
        return false;
    }
};

// jps_planner.cpp:54-95: walk the path once, dropping a vertex whenever the chord past it is clear and not longer
std::vector<V3> remove_corner_pts(const VoxelMap &map, const std::vector<V3> &path) {
    if (path.size() < 2) return path;
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<V3> out;
    V3 prev = path[0];
    out.push_back(prev);
    double cost1 = map.blocked(path[0], path[1]) ? inf : norm(path[0] - path[1]);
    for (size_t i = 1; i + 1 < path.size(); i++) {
        const V3 p1 = path[i], p2 = path[i + 1];
        const double cost2 = map.blocked(p1, p2) ? inf : norm(p1 - p2);
        const double cost3 = map.blocked(prev, p2) ? inf : norm(prev - p2);
        if (cost3 < cost1 + cost2)
            cost1 = cost3;
        else {
            out.push_back(p1);
            cost1 = norm(p1 - p2);
            prev = p1;
        }
    }
    out.push_back(path.back());
    return out;
}

// jps_planner.cpp:98-117: keep a vertex only where the step changes (second difference above 1e-2 in the 1-norm)
std::vector<V3> remove_line_pts(const std::vector<V3> &path) {
    if (path.size() < 3) return path;
    std::vector<V3> out;
    out.push_back(path.front());
    for (size_t i = 1; i + 1 < path.size(); i++) {
        const V3 a = path[i + 1] - path[i], b = path[i] - path[i - 1];
        const V3 p = a - b;
        if (std::fabs(p.x) + std::fabs(p.y) + std::fabs(p.z) > 1e-2) out.push_back(path[i]);
    }
    out.push_back(path.back());
    return out;
}

// jps_planner.cpp:118-148: the centres of the cells the polyline passes through, found by stepping 0.02 m along each edge
std::vector<V3> sample_path(const VoxelMap &map, const std::vector<V3> &path) {
    std::vector<V3> out;
    out.push_back(path.front());
    int last[3];
    map.to_cell(path.front(), last);
    for (size_t i = 0; i + 1 < path.size(); i++) {
        const V3 p1 = path[i], p2 = path[i + 1];
        const double l = norm(p2 - p1);
        // the reference's loop variable accumulates 0.02 (it is not i * 0.02) and the end point is sampled only when the
        // accumulation lands on it; its trailing `if (d < l)` can never fire
        for (double d = 0; d <= l; d += 0.02) {
            const double w1 = (l - d) / l, w2 = d / l;
            const V3 q = {w1 * p1.x + w2 * p2.x, w1 * p1.y + w2 * p2.y, w1 * p1.z + w2 * p2.z};
            int c[3];
            map.to_cell(q, c);
            if (c[0] != last[0] || c[1] != last[1] || c[2] != last[2]) {
                out.push_back(map.to_point(c));
                last[0] = c[0], last[1] = c[1], last[2] = c[2];
            }
        }
    }
    return out;
}

struct LegResult {
    int status = 0;  // JPSPlanner::status_: 0 ok, 1 start not free, 2 goal not free, -1 no path (jps_planner.cpp:341-389)
    std::vector<V3> raw, path, sample;
    int expanded = 0;
    int error = FRX_OK;
    std::string error_text;
};

// the occupancy array the search runs on (JPSPlanner::updateMap, jps_planner.cpp:309-323): occupied -> 1, everything else
// (free AND unknown) -> 0
std::vector<signed char> search_cells(const frx_voxel_map *m) {
    const size_t n = size_t(m->dim[0]) * m->dim[1] * m->dim[2];
    std::vector<signed char> c(n);
    for (size_t i = 0; i < n; i++) c[i] = m->cells[i] > 0 ? 1 : 0;
    return c;
}

void plan_leg(const frx_voxel_map *m, const signed char *cmap, V3 start, V3 goal, double eps, bool use_jps, LegResult &r) {
    VoxelMap map{m};
    int s[3], g[3];
    map.to_cell(start, s);
    if (!map.is_free(s)) {
        r.status = 1;
        return;
    }
    map.to_cell(goal, g);
    if (!map.is_free(g)) {
        r.status = 2;
        return;
    }
    GridSearch gs(cmap, m->dim[0], m->dim[1], m->dim[2], eps);
    if (!gs.allocated()) {
        r.error = FRX_ERR_ALLOC;
        r.error_text = "frx_jps_plan: no memory for the search slots";
        return;
    }
    const bool ok = gs.plan(s, g, use_jps, -1);
    r.expanded = gs.expanded();
    if (gs.broken_chain()) {
        r.error = FRX_ERR_INVALID_ARG;
        r.error_text = "frx_jps_plan: the parent chain of the goal does not reach the start (eps > 1 re-parented a closed cell); "
                       "the reference does not terminate on this input";
        return;
    }
    if (!ok || gs.path().empty()) {
        r.status = -1;
        return;
    }
    // raw path start -> goal at cell centres (jps_planner.cpp:393-406)
    const auto &p = gs.path();
    r.raw.reserve(p.size());
    for (size_t i = p.size(); i-- > 0;) {
        const Node &n = gs.node(p[i]);
        const int c[3] = {n.x, n.y, n.z};
        r.raw.push_back(map.to_point(c));
    }
    // simplify forwards, then backwards, then drop collinear vertices (jps_planner.cpp:408-416)
    std::vector<V3> q = remove_corner_pts(map, r.raw);
    std::reverse(q.begin(), q.end());
    q = remove_corner_pts(map, q);
    std::reverse(q.begin(), q.end());
    r.path = remove_line_pts(q);
    r.sample = sample_path(map, r.path);
}

int check_map(const frx_voxel_map *m, const char *who) {
    if (!m || !m->cells) return frx::set_error(FRX_ERR_INVALID_ARG, std::string(who) + ": null map");
    if (!(m->res > 0) || m->dim[0] <= 0 || m->dim[1] <= 0 || m->dim[2] <= 0)
        return frx::set_error(FRX_ERR_INVALID_ARG, std::string(who) + ": the map needs res > 0 and dim > 0 on all three axes");
    if (double(m->dim[0]) * m->dim[1] * m->dim[2] > 2.0e9)
        return frx::set_error(FRX_ERR_CAPACITY, std::string(who) + ": more than 2e9 cells (cell ids are 32-bit, as in the reference)");
    return FRX_OK;
}

int copy_out(const std::vector<V3> &v, int cap, int *n, double *out) {
    if (n) *n = int(v.size());
    if (!out) return FRX_OK;
    const int m = std::min<int>(cap, int(v.size()));
    for (int i = 0; i < m; i++) out[3 * i] = v[i].x, out[3 * i + 1] = v[i].y, out[3 * i + 2] = v[i].z;
    return int(v.size()) > cap ? FRX_ERR_CAPACITY : FRX_OK;
}

}  // namespace

extern "C" {

int frx_jps_tables(int *ns3, int *f13, int *f23, int *ns2, int *f12, int *f22) {
    const Neib3 &t3 = neib3();
    const Neib2 &t2 = neib2();
    // handed out in the reference's own storage order [id][axis][entry] (graph_search.h:96-98, :76-78)
    for (int id = 0; id < 27; id++)
        for (int a = 0; a < 3; a++) {
            for (int k = 0; k < 26; k++)
                if (ns3) ns3[(id * 3 + a) * 26 + k] = t3.ns[id][k][a];
            for (int k = 0; k < 12; k++) {
                if (f13) f13[(id * 3 + a) * 12 + k] = t3.f1[id][k][a];
                if (f23) f23[(id * 3 + a) * 12 + k] = t3.f2[id][k][a];
            }
        }
    for (int id = 0; id < 9; id++)
        for (int a = 0; a < 2; a++) {
            for (int k = 0; k < 8; k++)
                if (ns2) ns2[(id * 2 + a) * 8 + k] = t2.ns[id][k][a];
            for (int k = 0; k < 2; k++) {
                if (f12) f12[(id * 2 + a) * 2 + k] = t2.f1[id][k][a];
                if (f22) f22[(id * 2 + a) * 2 + k] = t2.f2[id][k][a];
            }
        }
    return FRX_OK;
}

int frx_grid_search(const signed char *cmap, const int *dim, const int *start, const int *goal, double eps, int use_jps,
                    int max_expand, int cap, int *n_path, int *path_xyz, int *n_expanded, double *cost) {
    if (!cmap || !dim || !start || !goal || !n_path)
        return frx::set_error(FRX_ERR_INVALID_ARG, "frx_grid_search: null argument");
    if (dim[0] <= 0 || dim[1] <= 0 || dim[2] < 0)
        return frx::set_error(FRX_ERR_INVALID_ARG, "frx_grid_search: dim must be positive (dim[2] = 0 selects the 2-D search)");
    const int Z = dim[2];
    if (double(dim[0]) * dim[1] * std::max(Z, 1) > 2.0e9)
        return frx::set_error(FRX_ERR_CAPACITY, "frx_grid_search: more than 2e9 cells");
    for (int i = 0; i < (Z ? 3 : 2); i++)
        if (start[i] < 0 || start[i] >= dim[i] || goal[i] < 0 || goal[i] >= dim[i])
            return frx::set_error(FRX_ERR_INVALID_ARG, "frx_grid_search: start or goal outside the grid (the reference indexes its "
                                                       "node table with them unchecked)");
    GridSearch gs(cmap, dim[0], dim[1], Z, eps);
    if (!gs.allocated()) return frx::set_error(FRX_ERR_ALLOC, "frx_grid_search: no memory for the search slots");
    const bool ok = gs.plan(start, goal, use_jps != 0, max_expand);
    if (n_expanded) *n_expanded = gs.expanded();
    if (gs.broken_chain())
        return frx::set_error(FRX_ERR_INVALID_ARG, "frx_grid_search: the parent chain of the goal does not reach the start "
                                                   "(inflated heuristic re-parented a closed cell)");
    const auto &p = gs.path();
    *n_path = ok ? int(p.size()) : 0;
    if (cost) *cost = ok ? gs.node(p.front()).g : -1.0;
    if (!ok) return FRX_OK;
    if (path_xyz)
        for (int i = 0; i < int(p.size()) && i < cap; i++) {
            const Node &n = gs.node(p[i]);
            path_xyz[3 * i] = n.x, path_xyz[3 * i + 1] = n.y, path_xyz[3 * i + 2] = n.z;
        }
    if (path_xyz && int(p.size()) > cap)
        return frx::set_error(FRX_ERR_CAPACITY, "frx_grid_search: path longer than cap");
    return FRX_OK;
}

int frx_map_mark_cloud(const double *origin, const int *dim, double res, int n_pts, const double *pts, signed char *cells) {
    if (!origin || !dim || !cells || (n_pts > 0 && !pts) || !(res > 0))
        return frx::set_error(FRX_ERR_INVALID_ARG, "frx_map_mark_cloud: null argument or res <= 0");
    frx_voxel_map m;
    for (int i = 0; i < 3; i++) m.origin[i] = origin[i], m.dim[i] = dim[i];
    m.res = res;
    m.cells = cells;
    VoxelMap map{&m};
    int marked = 0;
    for (int i = 0; i < n_pts; i++) {  // setObs with expand_size 0 (map_util.h:108-136): points outside are dropped
        int c[3];
        map.to_cell({pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, c);
        if (map.outside(c)) continue;
        cells[map.index(c)] = 100;
        marked++;
    }
    return marked;
}

int frx_map_is_blocked(const double *a, const double *b, void *map) {
    const frx_voxel_map *m = static_cast<const frx_voxel_map *>(map);
    VoxelMap vm{m};
    return vm.blocked({a[0], a[1], a[2]}, {b[0], b[1], b[2]}) ? 1 : 0;
}

int frx_jps_plan(const frx_voxel_map *map, const double *start, const double *goal, double eps, int use_jps, int cap, int *n_raw,
                 double *raw_path, int *n_path, double *path, int *n_sample, double *sample_path, int *status, int *n_expanded) {
    if (int rc = check_map(map, "frx_jps_plan")) return rc;
    if (!start || !goal || !status) return frx::set_error(FRX_ERR_INVALID_ARG, "frx_jps_plan: null argument");
    const std::vector<signed char> cmap = search_cells(map);
    LegResult r;
    plan_leg(map, cmap.data(), {start[0], start[1], start[2]}, {goal[0], goal[1], goal[2]}, eps, use_jps != 0, r);
    if (r.error != FRX_OK) return frx::set_error(r.error, r.error_text);
    *status = r.status;
    if (n_expanded) *n_expanded = r.expanded;
    int rc = copy_out(r.raw, cap, n_raw, raw_path);
    rc = std::min(rc, copy_out(r.path, cap, n_path, path));
    rc = std::min(rc, copy_out(r.sample, cap, n_sample, sample_path));
    if (rc != FRX_OK) return frx::set_error(rc, "frx_jps_plan: a path is longer than cap points");
    return FRX_OK;
}

int frx_route_plan(const frx_voxel_map *map, const double *start, const double *goal, int n_gates, const double *gates, double eps,
                   int use_jps, int n_threads, int cap, int *n_out, double *path_out, int *leg_status, int *leg_expanded) {
    if (int rc = check_map(map, "frx_route_plan")) return rc;
    if (!start || !goal || !n_out || n_gates < 0 || (n_gates > 0 && !gates))
        return frx::set_error(FRX_ERR_INVALID_ARG, "frx_route_plan: null argument");
    // legs start -> gate 0 -> ... -> gate n-1 -> goal (MinCoPlan_CPU.cpp:19-35).  They share nothing but the map, so they
    // run side by side; the reference runs them one after the other on one planner object.
    const int n_legs = n_gates + 1;
    std::vector<V3> wp;
    wp.push_back({start[0], start[1], start[2]});
    for (int i = 0; i < n_gates; i++) wp.push_back({gates[3 * i], gates[3 * i + 1], gates[3 * i + 2]});
    wp.push_back({goal[0], goal[1], goal[2]});
    const std::vector<signed char> cmap = search_cells(map);
    std::vector<LegResult> legs(n_legs);
    const int nt = std::max(1, std::min(n_threads > 0 ? n_threads : int(std::thread::hardware_concurrency()), n_legs));
    if (nt == 1) {
        for (int l = 0; l < n_legs; l++) plan_leg(map, cmap.data(), wp[l], wp[l + 1], eps, use_jps != 0, legs[l]);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++)
            pool.emplace_back([&, t] {
                for (int l = t; l < n_legs; l += nt) plan_leg(map, cmap.data(), wp[l], wp[l + 1], eps, use_jps != 0, legs[l]);
            });
        for (auto &th : pool) th.join();
    }
    std::vector<V3> out;
    bool all_ok = true;
    for (int l = 0; l < n_legs; l++) {
        if (legs[l].error != FRX_OK) return frx::set_error(legs[l].error, legs[l].error_text);
        if (leg_status) leg_status[l] = legs[l].status;
        if (leg_expanded) leg_expanded[l] = legs[l].expanded;
        if (legs[l].status != 0) {
            all_ok = false;
            continue;
        }
        // the reference does not look at the legs' verdicts and splices whatever getSamplePath() holds; a failed leg leaves
        // the previous leg's samples there.  Here a failed leg is reported and contributes nothing.
        if (l > 0 && !out.empty()) out.pop_back();
        out.insert(out.end(), legs[l].sample.begin(), legs[l].sample.end());
    }
    if (!all_ok) {
        *n_out = 0;
        return FRX_OK;
    }
    const int rc = copy_out(out, cap, n_out, path_out);
    if (rc != FRX_OK) return frx::set_error(rc, "frx_route_plan: the route is longer than cap points");
    return FRX_OK;
}
}
