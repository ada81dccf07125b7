// Compile-and-link check of the C++ mirror (include/se3gcopter_amd.hpp) against libfrx.so; also runs on a CPU-only
// box, where setup() must return false with the "no HIP device" diagnostic (no silent fallback).
#include <cstdio>
#include <cstring>
#include "../include/se3gcopter_amd.hpp"

int main() {
    frx_amd::SE3GCOPTER opt;
    frx_amd::SE3GCOPTER::Candidate c;
    std::memset(c.iniState, 0, sizeof(c.iniState)); std::memset(c.finState, 0, sizeof(c.finState));
    c.finState[1] = 4.0;
    frx_amd::Polytope cell;                       // axis-aligned box [-4,4] x [-4,8] x [0,3]
    const double n[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    const double p[6][3] = {{4, 0, 0}, {-4, 0, 0}, {0, 8, 0}, {0, -4, 0}, {0, 0, 3}, {0, 0, 0}};
    for (int k = 0; k < 6; k++) { for (int d = 0; d < 3; d++) cell.h.push_back(n[k][d]); for (int d = 0; d < 3; d++) cell.h.push_back(p[k][d]); }
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int e = 0; e < 2; e++) { cell.v.push_back(a ? 4 : -4); cell.v.push_back(b ? 8 : -4); cell.v.push_back(e ? 3 : 0); }
    c.cells.push_back(cell);
    const double w[4] = {1e7, 1e4, 1e4, 1e4};
    bool ok = opt.setup(1000.0, 0.0, {c}, INFINITY, 8, 0.5, 0.15, 0.08, 14.0, 5.0, 12.0, 3.8, 9.81, w, true);
    if (!ok) { std::printf("setup failed: %s\n", opt.last_error().c_str()); return frx_device_count() > 0 ? 1 : 0; }
    std::vector<std::vector<frx_amd::PieceOut>> trajs;
    double jc = opt.optimize(trajs, 1e-6);
    std::printf("jerk cost %.6f, %zu pieces, duration %.4f, status %d\n", jc, trajs[0].size(), trajs[0][0].duration, opt.status()[0]);
    return (trajs[0].size() == 1 && std::isfinite(jc) && opt.status()[0] >= 0) ? 0 : 1;
}
