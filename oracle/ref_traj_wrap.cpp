// TEST INFRASTRUCTURE.  Compiles the reference's result type (trajectory.hpp: Piece / Trajectory) together with its root
// finder (root_finder.hpp) UNMODIFIED from where they lie under /root/reference, against oracle/eigen_shim, into
// oracle/_ref/libref_traj.so.  Pins the product's trajectory evaluation, wire format and max-rate post-checks
// (frx_traj_to_msg, frx_msg_sample, frx_traj_max_rates); never used by the product.
#include <cstring>
#include <set>
#include <cfloat>
#include <cmath>
#include <iostream>
#include <vector>
#include <Eigen/Eigen>
#include "trajectory.hpp"

namespace {
Piece make_piece(double dur, const double *c18) {          // c18: [axis][column], column j = power 5-j (Piece's own layout)
    Eigen::Matrix<double, 3, 6> cm;
    for (int d = 0; d < 3; d++) for (int j = 0; j < 6; j++) cm(d, j) = c18[6 * d + j];
    return Piece(dur, cm);
}
Trajectory make_traj(int n, const double *durs, const double *c18) {
    Trajectory t;
    for (int i = 0; i < n; i++) t.emplace_back(make_piece(durs[i], c18 + 18 * i));
    return t;
}
void put(const Eigen::Vector3d &v, double *o) { o[0] = v(0); o[1] = v(1); o[2] = v(2); }
} // namespace

extern "C" {

// Piece::getMaxVelRate / getMaxAccRate (trajectory.hpp:177-273)
void ref_piece_max_rates(double dur, const double *c18, double *out2) {
    const Piece p = make_piece(dur, c18);
    out2[0] = p.getMaxVelRate();
    out2[1] = p.getMaxAccRate();
}
// Trajectory::getPos/getVel/getAcc/getJer at time t (trajectory.hpp:453-475), getTotalDuration, getMaxVelRate/getMaxAccRate
void ref_traj_eval(int n, const double *durs, const double *c18, double t, double *pos, double *vel, double *acc, double *jer) {
    const Trajectory tr = make_traj(n, durs, c18);
    put(tr.getPos(t), pos); put(tr.getVel(t), vel); put(tr.getAcc(t), acc); put(tr.getJer(t), jer);
}
void ref_traj_max_rates(int n, const double *durs, const double *c18, double *out3) {
    const Trajectory tr = make_traj(n, durs, c18);
    out3[0] = tr.getMaxVelRate(); out3[1] = tr.getMaxAccRate(); out3[2] = tr.getTotalDuration();
}
// Piece::normalizePosCoeffMat (trajectory.hpp:131-141), [axis][column]
void ref_piece_normalized(double dur, const double *c18, double *out18) {
    const auto m = make_piece(dur, c18).normalizePosCoeffMat();
    for (int d = 0; d < 3; d++) for (int j = 0; j < 6; j++) out18[6 * d + j] = m(d, j);
}
}
