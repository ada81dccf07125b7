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
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "frx.h"

namespace frx_amd {

struct Polytope {                 // one corridor cell
    std::vector<double> h;        // 6 x K column-major: column = (outer normal, point)   (MinCoPlan_CPU.cpp:93-105)
    std::vector<double> v;        // 3 x nv column-major vertices (output of geoutils::enumerateVs)
};

struct PieceOut { double duration; double coeff[3][6]; };   // Piece(T, 3x6 highest power first) as getTraj emits (CPU.hpp:561)

class SE3GCOPTER {
public:
    ~SE3GCOPTER() { kill_kernel(); }

    // One candidate = (iniState 3x3 col-major, finState, cells[coarseN], overlaps[coarseN-1]).
    struct Candidate {
        double iniState[9], finState[9];
        std::vector<Polytope> cells;
        std::vector<std::vector<double>> overlap_vertices;   // vertices of cell i ∩ cell i+1 (3 x nv col-major)
    };

    // Argument order and meaning of SE3GCOPTER::setup (CPU.hpp:1076-1092); returns false like the reference when a
    // polytope has an empty interior, and also when no MI355X is usable (last_error() says which).
    bool setup(double rh, double st, const std::vector<Candidate> &cands, double gridRes, int itgSpaces, double horiHalfLen,
               double vertHalfLen, double margin, double vm, double minThrAcc, double maxThrAcc, double bodyRateMax, double g,
               const double w[4], bool c2diffeo, int device = 0) {
        kill_kernel();
        frx_config cfg;
        cfg.rho = rh; cfg.total_t = st; cfg.grid_res = gridRes; cfg.qd_intervals = itgSpaces; cfg.c2_diffeo = c2diffeo ? 1 : 0;
        cfg.horiz_half_len = horiHalfLen; cfg.vert_half_len = vertHalfLen; cfg.safe_margin = margin; cfg.vel_max = vm;
        cfg.thr_acc_min = minThrAcc; cfg.thr_acc_max = maxThrAcc; cfg.body_rate_max = bodyRateMax; cfg.grav_acc = g;
        for (int i = 0; i < 4; i++) cfg.penalty_pvtb[i] = w[i];
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
        int rc = frx_problem_create(&cfg, device, (int)cands.size(), coarse.data(), ini.data(), fin.data(), hoff.data(), hrec.data(),
                                    voff.data(), vrec.data(), &p_);
        if (rc != FRX_OK) { err_ = frx_last_error(); p_ = nullptr; return false; }
        int t[6];
        frx_problem_totals(p_, t);
        B_ = t[0]; P_ = t[1]; NX_ = t[3];
        poff_.resize(B_ + 1);
        frx_problem_layout(p_, poff_.data(), nullptr, nullptr, nullptr);
        return true;
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
    frx_problem *p_ = nullptr;
    int B_ = 0, P_ = 0, NX_ = 0;
    std::vector<int> poff_, status_, iters_, evals_;
    std::vector<double> jerk_;
    std::string err_;
};

} // namespace frx_amd
