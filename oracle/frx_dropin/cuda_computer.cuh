// TEST INFRASTRUCTURE / integration artefact.  The drop-in for the reference's `class cuda_computer`
// (/root/reference/src/plan_manage/include/cuda_computer.cuh:100-140; implementation src/cuda_computer.cu:413-579) with the reference's OWN
// signatures, backed by libfrx.so: with this directory in front of the reference's include path, se3gcopter_gpu.hpp compiles UNMODIFIED
// (`#include <cuda_computer.cuh>`, se3gcopter_gpu.hpp:29; member `cuda_computer paraller`, :45; call site :219-227; setup :1015; kill_kernel :907-909)
// and MINCO_S3::addTimeIntPenalty runs on the MI355X through frx_penalty_eval - the literal "behind the existing plan_manage API".
// oracle/ref_gcopter_gpu_wrap.cpp builds exactly that; tests/test_reference_gpu_header.py runs it.
//
// What differs from the reference's class, and why:
//   * the reference launches a persistent kernel in setup() and feeds it ~257 KB through mapped memory on EVERY compute() (cc.cu:492-527).  Here the
//     polytopes and parameters go to the device once, on the first compute() after a setup() (they only arrive with compute()'s arguments), and
//     later calls send (T, b) only; a call whose constant arguments differ from the uploaded ones re-creates the handle.
//   * cc.cu:97 indexes the polytope of piece i as cfgHs[i] instead of cfgHs[idxHs(i)]; this class follows the CPU path (idxHs), which is the oracle.
//   * errors surface as std::runtime_error with frx_last_error()'s text (the reference throws CudaException on allocation failure only and checks nothing else).
#ifndef FRX_DROPIN_CUDA_COMPUTER_CUH
#define FRX_DROPIN_CUDA_COMPUTER_CUH

#include <Eigen/Eigen>

#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "frx.h"

class cuda_computer {
public:
    cuda_computer() {}
    ~cuda_computer() { kill_kernel(); }
    cuda_computer(const cuda_computer &) = delete;
    cuda_computer &operator=(const cuda_computer &) = delete;

    int segs = 0;
    long n_compute = 0;                                     // calls served (the reference's GPU header calls compute TWICE per evaluation, se3gcopter_gpu.hpp:219-227)

    // cuda_computer::setup (cc.cu:413-466): the reference allocates its mapped buffers and launches the persistent kernel; a new plan begins
    void setup(const int pieceNum) {
        kill_kernel();
        segs = pieceNum;
    }

    // cuda_computer::compute (cc.cu:469-563): ACCUMULATES the penalty into cost / gdT / gdC (cc.cu:551-558)
    void compute(const Eigen::VectorXi cons, const Eigen::VectorXi &idxHs, const std::vector<Eigen::MatrixXd> &cfgHs, const Eigen::Vector3d &ellipsoid,
                 const double safeMargin, const double vMax, const double thrAccMin, const double thrAccMax, const double bdrMax, const double gAcc,
                 const Eigen::Vector4d ci, double &cost, Eigen::VectorXd &gdT, Eigen::MatrixXd &gdC, int pieceNum, Eigen::VectorXd T1, Eigen::MatrixXd b) {
        const int N = pieceNum;
        // the constant arguments as one key: {kappa, limits, weights, idxHs, every polytope}
        std::vector<double> key;
        key.reserve(16 + (size_t)N);
        const int kappa = cons(0);
        for (int i = 0; i < N; i++) if (cons(i) != kappa) throw std::runtime_error("cuda_computer drop-in: one quadrature resolution per plan (cons is set with setConstant, se3gcopter_gpu.hpp:964)");
        const double scal[13] = {(double)N, (double)kappa, ellipsoid(0), ellipsoid(1), ellipsoid(2), safeMargin, vMax, thrAccMin, thrAccMax, bdrMax, gAcc, ci(0), ci(1)};
        key.insert(key.end(), scal, scal + 13); key.push_back(ci(2)); key.push_back(ci(3));
        for (int i = 0; i < N; i++) key.push_back((double)idxHs(i));
        std::vector<int> hoff(1, 0);
        std::vector<double> hrec;
        for (size_t m = 0; m < cfgHs.size(); m++) {
            const int K = (int)cfgHs[m].cols();
            for (int k = 0; k < K; k++) for (int r = 0; r < 6; r++) hrec.push_back(cfgHs[m](r, k));
            hoff.push_back(hoff.back() + K);
        }
        key.insert(key.end(), hrec.begin(), hrec.end());
        if (!handle || key != uploaded) {
            kill_kernel();
            frx_config cfg;
            cfg.rho = 1.0; cfg.total_t = 0.0; cfg.grid_res = INFINITY; cfg.qd_intervals = kappa; cfg.c2_diffeo = 1;
            if (ellipsoid(0) != ellipsoid(1)) throw std::runtime_error("cuda_computer drop-in: the ellipsoid is (horiz, horiz, vert) (se3gcopter_gpu.hpp:987-989)");
            cfg.horiz_half_len = ellipsoid(0); cfg.vert_half_len = ellipsoid(2); cfg.safe_margin = safeMargin;
            cfg.vel_max = vMax; cfg.thr_acc_min = thrAccMin; cfg.thr_acc_max = thrAccMax; cfg.body_rate_max = bdrMax; cfg.grav_acc = gAcc;
            for (int q = 0; q < 4; q++) cfg.penalty_pvtb[q] = ci(q);
            std::vector<int> poly(N);
            for (int i = 0; i < N; i++) poly[i] = idxHs(i);
            if (frx_penalty_problem_create(&cfg, 0, 1, &N, poly.data(), hoff.data(), hrec.data(), &handle) != FRX_OK) { handle = nullptr; throw std::runtime_error(frx_last_error()); }
            uploaded.swap(key);
            Tbuf.resize(N); Cbuf.resize((size_t)18 * N); gTbuf.resize(N); gCbuf.resize((size_t)18 * N);
        }
        // frx is piece-major: row 6 i + k of b = 3 contiguous doubles of piece i's block; Eigen's MatrixXd is column-major
        for (int i = 0; i < N; i++) Tbuf[i] = T1(i);
        for (int r = 0; r < 6 * N; r++) for (int d = 0; d < 3; d++) { Cbuf[(size_t)3 * r + d] = b(r, d); gCbuf[(size_t)3 * r + d] = gdC(r, d); }
        for (int i = 0; i < N; i++) gTbuf[i] = gdT(i);
        if (frx_penalty_eval(handle, Tbuf.data(), Cbuf.data(), &cost, gTbuf.data(), gCbuf.data()) != FRX_OK) throw std::runtime_error(frx_last_error());
        for (int i = 0; i < N; i++) gdT(i) = gTbuf[i];
        for (int r = 0; r < 6 * N; r++) for (int d = 0; d < 3; d++) gdC(r, d) = gCbuf[(size_t)3 * r + d];
        n_compute++;
    }

    // cuda_computer::kill_kernel (cc.cu:44-49): ends the persistent kernel; here: frees the handle
    void kill_kernel() {
        if (handle) frx_problem_destroy(handle);
        handle = nullptr;
        uploaded.clear();
    }

private:
    frx_problem *handle = nullptr;
    std::vector<double> uploaded, Tbuf, Cbuf, gTbuf, gCbuf;
};

#endif
