// se3gcopter_amd.hpp — header-only C++ mirror of the reference's operator interface on top of the C ABI (frx.h).
//
// Same names, argument meaning and error behaviour as
//   SE3GCOPTER::setup    (src/plan_manage/include/se3gcopter/se3gcopter_cpu.hpp:1076-1186)   -> bool, false on empty polytope
//   SE3GCOPTER::optimize (se3gcopter_cpu.hpp:1230-1268)                                       -> returns the jerk cost
//   cuda_computer::{setup,compute,kill_kernel} (src/plan_manage/include/cuda_computer.cuh:100-140)
// but with plain arrays instead of Eigen types (Eigen is not available in this build environment; INTEGRATION.md
// shows the two-line Eigen adapters a maintainer adds on the reference side).  A batch of B candidates is the native
// unit; B = 1 reproduces the reference's single-trajectory call.
#pragma once
#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "frx.h"

namespace frx_amd {

struct Polytope {                 // one corridor cell
    std::vector<double> h;        // 6 x K column-major: column = (outer normal, point)   (MinCoPlan_CPU.cpp:93-105)
    std::vector<double> v;        // 3 x nv column-major vertices (output of geoutils::enumerateVs)
};

struct PieceOut { double duration; double coeff[3][6]; };   // Piece(T, 3x6 highest power first) as getTraj emits (CPU.hpp:561)

// Piece / Trajectory: the evaluation interface of the reference's result type (trajectory.hpp:43-145, 317-491) on top of the
// pieces optimize() returns; same member names, plain arrays instead of Eigen vectors.
class Piece {
public:
    Piece() { po_.duration = 0.0; for (auto &r : po_.coeff) for (double &c : r) c = 0.0; }
    explicit Piece(const PieceOut &po) : po_(po) {}
    int getDim() const { return 3; }                                   // trajectory.hpp:55-58
    int getOrder() const { return 5; }                                 // :60-63
    double getDuration() const { return po_.duration; }                // :65-68
    const double (&getCoeffMat() const)[3][6] { return po_.coeff; }    // :70-73 (row = axis, column j = power 5-j)
    void getPos(double t, double *out) const { eval(t, 0, out); }      // :75-85
    void getVel(double t, double *out) const { eval(t, 1, out); }      // :87-99
    void getAcc(double t, double *out) const { eval(t, 2, out); }      // :101-115
    void getJer(double t, double *out) const { eval(t, 3, out); }      // :117-129
    // getMaxVelRate / getMaxAccRate (:177-273) through the library's post-check (frx_traj_max_rates)
    double getMaxVelRate() const { double v = 0.0; rates(&v, nullptr); return v; }
    double getMaxAccRate() const { double a = 0.0; rates(nullptr, &a); return a; }
    // normalizePosCoeffMat (:131-141): column j scaled by duration^(5-j)
    void normalizePosCoeffMat(double (&out)[3][6]) const {
        double t = 1.0;
        for (int j = 5; j >= 0; j--) { for (int d = 0; d < 3; d++) out[d][j] = po_.coeff[d][j] * t; t *= po_.duration; }
    }
private:
    void rates(double *v, double *a) const {
        double c[18];                                                  // frx layout: row = power, column = axis
        for (int k = 0; k < 6; k++) for (int d = 0; d < 3; d++) c[3 * k + d] = po_.coeff[d][5 - k];
        if (frx_traj_max_rates(1, &po_.duration, c, v, a) != FRX_OK) throw std::runtime_error(frx_last_error());
    }
    void eval(double t, int der, double *out) const {                  // lowest power first, as the reference accumulates
        static const double fac[4][6] = {{1, 1, 1, 1, 1, 1}, {0, 1, 2, 3, 4, 5}, {0, 0, 2, 6, 12, 20}, {0, 0, 0, 6, 24, 60}};
        out[0] = out[1] = out[2] = 0.0;
        double tn = 1.0;
        for (int k = der; k <= 5; k++) {                               // power k sits in column 5-k
            for (int d = 0; d < 3; d++) out[d] += fac[der][k] * tn * po_.coeff[d][5 - k];
            tn *= t;
        }
    }
    PieceOut po_;
};

class Trajectory {
public:
    Trajectory() {}
    explicit Trajectory(const std::vector<PieceOut> &ps) { for (const PieceOut &p : ps) pieces_.push_back(Piece(p)); }
    int getPieceNum() const { return (int)pieces_.size(); }            // trajectory.hpp:337-340
    std::vector<double> getDurations() const { std::vector<double> d; for (const Piece &p : pieces_) d.push_back(p.getDuration()); return d; }   // :342-351
    double getTotalDuration() const { double s = 0.0; for (const Piece &p : pieces_) s += p.getDuration(); return s; }                          // :353-362
    const Piece &operator[](int i) const { return pieces_[i]; }        // :385-388
    // locatePieceIdx (:432-451): t becomes the local time of the returned piece; times past the end land at the end of the last piece
    int locatePieceIdx(double &t) const {
        const int N = getPieceNum();
        int idx;
        double dur;
        for (idx = 0; idx < N && t > (dur = pieces_[idx].getDuration()); idx++) t -= dur;
        if (idx == N) { idx--; t += pieces_[idx].getDuration(); }
        return idx;
    }
    void getPos(double t, double *out) const { const int i = locatePieceIdx(t); pieces_[i].getPos(t, out); }   // :453-457
    void getVel(double t, double *out) const { const int i = locatePieceIdx(t); pieces_[i].getVel(t, out); }   // :459-463
    void getAcc(double t, double *out) const { const int i = locatePieceIdx(t); pieces_[i].getAcc(t, out); }   // :465-469
    void getJer(double t, double *out) const { const int i = locatePieceIdx(t); pieces_[i].getJer(t, out); }   // :471-475
    // junction position between pieces i-1 and i, i = 0..N (getJuncPos, :477-491)
    void getJuncPos(int i, double *out) const { if (i != getPieceNum()) pieces_[i].getPos(0.0, out); else pieces_[i - 1].getPos(pieces_[i - 1].getDuration(), out); }
    double getMaxVelRate() const { double m = 0.0; for (const Piece &p : pieces_) m = std::max(m, p.getMaxVelRate()); return m; }   // :505-518
    double getMaxAccRate() const { double m = 0.0; for (const Piece &p : pieces_) m = std::max(m, p.getMaxAccRate()); return m; }   // :520-533
    // MavGlobalPlanner::traj2msg (se3_planner.cpp:31-58): the array fields of quadrotor_msgs/PolynomialTrajectory
    struct Msg { std::vector<double> coef_x, coef_y, coef_z, time; std::vector<unsigned> order; unsigned num_order = 5, num_segment = 0; double mag_coeff = 1.0; };
    Msg toMsg() const {
        Msg m;
        for (const Piece &p : pieces_) {
            double c[3][6];
            p.normalizePosCoeffMat(c);
            for (int j = 0; j < 6; j++) { m.coef_x.push_back(c[0][j]); m.coef_y.push_back(c[1][j]); m.coef_z.push_back(c[2][j]); }
            m.time.push_back(p.getDuration()); m.order.push_back(5);
        }
        m.num_segment = (unsigned)pieces_.size();
        return m;
    }
private:
    std::vector<Piece> pieces_;
};

class SE3GCOPTER {
public:
    SE3GCOPTER() = default;
    ~SE3GCOPTER() { kill_kernel(); }
    // owns the device handle: movable, not copyable (a copy would destroy the same handle twice)
    SE3GCOPTER(const SE3GCOPTER &) = delete;
    SE3GCOPTER &operator=(const SE3GCOPTER &) = delete;
    SE3GCOPTER(SE3GCOPTER &&o) noexcept { *this = std::move(o); }
    SE3GCOPTER &operator=(SE3GCOPTER &&o) noexcept;

    // One candidate = (iniState 3x3 col-major, finState, cells[coarseN], overlaps[coarseN-1]).
    struct Candidate {
        double iniState[9], finState[9];
        std::vector<Polytope> cells;
        std::vector<std::vector<double>> overlap_vertices;   // vertices of cell i ∩ cell i+1 (3 x nv col-major)
    };

    // SE3GCOPTER::setup exactly as the reference declares it (CPU.hpp:1076-1092), for ONE trajectory: the caller passes the
    // H-polytopes only (hPolys[i] = 6 x K_i column-major) and the vertices are enumerated inside, as extractVs does
    // (CPU.hpp:1031-1074).  Returns false when a cell or an overlap has no interior, or when no MI355X is usable.
    bool setup(double rh, double st, const double iniState[9], const double finState[9], const std::vector<std::vector<double>> &hPolys,
               double gridRes, int itgSpaces, double horiHalfLen, double vertHalfLen, double margin, double vm, double minThrAcc,
               double maxThrAcc, double bodyRateMax, double g, const double w[4], bool c2diffeo, int device = 0) {
        HCandidate c;
        for (int i = 0; i < 9; i++) { c.iniState[i] = iniState[i]; c.finState[i] = finState[i]; }
        c.hPolys = hPolys;
        return setup(rh, st, std::vector<HCandidate>{c}, gridRes, itgSpaces, horiHalfLen, vertHalfLen, margin, vm, minThrAcc, maxThrAcc, bodyRateMax, g, w, c2diffeo, device);
    }
    // the same for a batch of candidates, each with its own corridor
    struct HCandidate { double iniState[9], finState[9]; std::vector<std::vector<double>> hPolys; };
    bool setup(double rh, double st, const std::vector<HCandidate> &cands, double gridRes, int itgSpaces, double horiHalfLen,
               double vertHalfLen, double margin, double vm, double minThrAcc, double maxThrAcc, double bodyRateMax, double g,
               const double w[4], bool c2diffeo, int device = 0) {
        kill_kernel();
        const frx_config cfg = make_config(rh, st, gridRes, itgSpaces, horiHalfLen, vertHalfLen, margin, vm, minThrAcc, maxThrAcc, bodyRateMax, g, w, c2diffeo);
        std::vector<int> coarse, hoff{0};
        std::vector<double> ini, fin, hrec;
        for (const HCandidate &c : cands) {
            coarse.push_back((int)c.hPolys.size());
            ini.insert(ini.end(), c.iniState, c.iniState + 9);
            fin.insert(fin.end(), c.finState, c.finState + 9);
            for (const auto &h : c.hPolys) { hoff.push_back(hoff.back() + (int)(h.size() / 6)); hrec.insert(hrec.end(), h.begin(), h.end()); }
        }
        return finish_setup(frx_problem_create_from_h(&cfg, device, (int)cands.size(), coarse.data(), ini.data(), fin.data(), hoff.data(), hrec.data(), &p_));
    }

    // Variant with caller-supplied vertices (e.g. the reference's own enumerateVs output, to keep its vertex order).
    bool setup(double rh, double st, const std::vector<Candidate> &cands, double gridRes, int itgSpaces, double horiHalfLen,
               double vertHalfLen, double margin, double vm, double minThrAcc, double maxThrAcc, double bodyRateMax, double g,
               const double w[4], bool c2diffeo, int device = 0) {
        kill_kernel();
        const frx_config cfg = make_config(rh, st, gridRes, itgSpaces, horiHalfLen, vertHalfLen, margin, vm, minThrAcc, maxThrAcc, bodyRateMax, g, w, c2diffeo);
        std::vector<int> coarse, hoff{0}, voff{0};
        std::vector<double> ini, fin, hrec, vrec;
        for (const Candidate &c : cands) {
            const int cN = (int)c.cells.size();
            if ((int)c.overlap_vertices.size() != cN - 1) { err_ = "need coarseN-1 overlap polytopes"; return false; }
            coarse.push_back(cN);
            ini.insert(ini.end(), c.iniState, c.iniState + 9);
            fin.insert(fin.end(), c.finState, c.finState + 9);
            for (int i = 0; i < cN; i++) {
                hoff.push_back(hoff.back() + (int)(c.cells[i].h.size() / 6));
                hrec.insert(hrec.end(), c.cells[i].h.begin(), c.cells[i].h.end());
                voff.push_back(voff.back() + (int)(c.cells[i].v.size() / 3));
                vrec.insert(vrec.end(), c.cells[i].v.begin(), c.cells[i].v.end());
                if (i + 1 < cN) {
                    voff.push_back(voff.back() + (int)(c.overlap_vertices[i].size() / 3));
                    vrec.insert(vrec.end(), c.overlap_vertices[i].begin(), c.overlap_vertices[i].end());
                }
            }
        }
        return finish_setup(frx_problem_create(&cfg, device, (int)cands.size(), coarse.data(), ini.data(), fin.data(), hoff.data(), hrec.data(),
                                               voff.data(), vrec.data(), &p_));
    }

    // SE3GCOPTER::optimize(Trajectory &traj, const double &relCostTol) (CPU.hpp:1230) for the single-trajectory setup
    double optimize(Trajectory &traj, const double &relCostTol) {
        std::vector<std::vector<PieceOut>> trajs;
        const double jc = optimize(trajs, relCostTol);
        traj = Trajectory(trajs[0]);
        return jc;
    }

    // SE3GCOPTER::optimize (CPU.hpp:1230): fills one trajectory per candidate, returns the jerk cost of candidate 0
    // (all of them in jerk_costs()).  The L-BFGS status the reference throws away is kept in status().
    double optimize(std::vector<std::vector<PieceOut>> &trajs, const double &relCostTol) {
        if (!p_) throw std::runtime_error("SE3GCOPTER::optimize before a successful setup");
        std::vector<double> x(NX_), C((size_t)P_ * 18), T(P_), obj(B_);
        jerk_.assign(B_, 0.0); status_.assign(B_, 0); iters_.assign(B_, 0); evals_.assign(B_, 0);
        if (frx_initial_guess(p_, x.data()) != FRX_OK) throw std::runtime_error(frx_last_error());
        frx_lbfgs_params pm;
        frx_lbfgs_gcopter_params(&pm, relCostTol);
        if (frx_optimize(p_, &pm, x.data(), C.data(), T.data(), jerk_.data(), obj.data(), status_.data(), iters_.data(), evals_.data()) != FRX_OK)
            throw std::runtime_error(frx_last_error());
        trajs.assign(B_, {});
        for (int b = 0; b < B_; b++)
            for (int gp = poff_[b]; gp < poff_[b + 1]; gp++) {
                PieceOut po;
                po.duration = T[gp];
                for (int d = 0; d < 3; d++)
                    for (int k = 0; k < 6; k++) po.coeff[d][k] = C[(size_t)gp * 18 + (5 - k) * 3 + d];   // rowwise().reverse(), CPU.hpp:561
                trajs[b].push_back(po);
            }
        return jerk_[0];
    }

    // cuda_computer::compute (cc.cuh:118-134): accumulates the penalty of the whole batch into cost/gdT/gdC.
    void compute(const double *T1, const double *b, double *cost, double *gdT, double *gdC) {
        if (!p_) throw std::runtime_error("compute before setup");
        if (frx_penalty_eval(p_, T1, b, cost, gdT, gdC) != FRX_OK) throw std::runtime_error(frx_last_error());
    }
    void kill_kernel() { if (p_) { frx_problem_destroy(p_); p_ = nullptr; } }   // GPU.hpp:907-909

    const std::vector<double> &jerk_costs() const { return jerk_; }
    const std::vector<int> &status() const { return status_; }
    const std::vector<int> &iterations() const { return iters_; }
    const std::string &last_error() const { return err_; }
    frx_problem *handle() { return p_; }

private:
    static frx_config make_config(double rh, double st, double gridRes, int itgSpaces, double horiHalfLen, double vertHalfLen, double margin, double vm,
                                  double minThrAcc, double maxThrAcc, double bodyRateMax, double g, const double w[4], bool c2diffeo) {
        frx_config cfg;
        cfg.rho = rh; cfg.total_t = st; cfg.grid_res = gridRes; cfg.qd_intervals = itgSpaces; cfg.c2_diffeo = c2diffeo ? 1 : 0;
        cfg.horiz_half_len = horiHalfLen; cfg.vert_half_len = vertHalfLen; cfg.safe_margin = margin; cfg.vel_max = vm;
        cfg.thr_acc_min = minThrAcc; cfg.thr_acc_max = maxThrAcc; cfg.body_rate_max = bodyRateMax; cfg.grav_acc = g;
        for (int i = 0; i < 4; i++) cfg.penalty_pvtb[i] = w[i];
        return cfg;
    }
    bool finish_setup(int rc) {
        if (rc != FRX_OK) { err_ = frx_last_error(); p_ = nullptr; return false; }
        int t[6];
        frx_problem_totals(p_, t);
        B_ = t[0]; P_ = t[1]; NX_ = t[3];
        poff_.resize(B_ + 1);
        frx_problem_layout(p_, poff_.data(), nullptr, nullptr, nullptr);
        return true;
    }
    frx_problem *p_ = nullptr;
    int B_ = 0, P_ = 0, NX_ = 0;
    std::vector<int> poff_, status_, iters_, evals_;
    std::vector<double> jerk_;
    std::string err_;
};

inline SE3GCOPTER &SE3GCOPTER::operator=(SE3GCOPTER &&o) noexcept {
    if (this != &o) {
        kill_kernel();
        p_ = o.p_; o.p_ = nullptr;
        B_ = o.B_; P_ = o.P_; NX_ = o.NX_;
        poff_ = std::move(o.poff_); status_ = std::move(o.status_); iters_ = std::move(o.iters_); evals_ = std::move(o.evals_);
        jerk_ = std::move(o.jerk_); err_ = std::move(o.err_);
    }
    return *this;
}

} // namespace frx_amd
