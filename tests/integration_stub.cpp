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
    if (!(trajs[0].size() == 1 && std::isfinite(jc) && opt.status()[0] >= 0)) return 1;

    // the reference's own call shape (MinCoPlan_CPU.cpp:114-124): H-polytopes only, a Trajectory back; two cells
    frx_amd::SE3GCOPTER opt2;
    std::vector<std::vector<double>> hPolys(2);
    for (int cellI = 0; cellI < 2; cellI++)
        for (int k = 0; k < 6; k++) {
            for (int d = 0; d < 3; d++) hPolys[cellI].push_back(n[k][d]);
            for (int d = 0; d < 3; d++) hPolys[cellI].push_back(p[k][d] + (d == 1 ? 6.0 * cellI : 0.0));      // second box shifted 6 m along y
        }
    double fin2[9] = {0, 10, 1, 0, 0, 0, 0, 0, 0}, ini2[9] = {0, 0, 1, 0, 0, 0, 0, 0, 0};
    if (!opt2.setup(1000.0, 0.0, ini2, fin2, hPolys, INFINITY, 8, 0.5, 0.15, 0.08, 14.0, 5.0, 12.0, 3.8, 9.81, w, true)) { std::printf("setup(H) failed: %s\n", opt2.last_error().c_str()); return 1; }
    frx_amd::Trajectory traj;
    const double jc2 = opt2.optimize(traj, 1e-6);
    double p0[3], pe[3], ve[3];
    traj.getPos(0.0, p0); traj.getPos(traj.getTotalDuration(), pe); traj.getVel(traj.getTotalDuration(), ve);
    const frx_amd::Trajectory::Msg msg = traj.toMsg();
    std::printf("H-only setup: jerk cost %.6f, %d pieces, total %.4f s, end (%.6f %.6f %.6f), msg %u segments, max vel %.3f acc %.3f\n", jc2, traj.getPieceNum(),
                traj.getTotalDuration(), pe[0], pe[1], pe[2], msg.num_segment, traj.getMaxVelRate(), traj.getMaxAccRate());   // MinCoPlan_CPU.cpp:131-132
    const bool ends = std::fabs(p0[1]) < 1e-9 && std::fabs(pe[1] - 10.0) < 1e-9 && std::fabs(pe[2] - 1.0) < 1e-9 && std::fabs(ve[1]) < 1e-9;
    return (traj.getPieceNum() == 2 && ends && msg.coef_x.size() == 12 && std::isfinite(jc2)) ? 0 : 1;
}
