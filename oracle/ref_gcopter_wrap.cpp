// TEST INFRASTRUCTURE — builds oracle/_ref/libref_gcopter.so from the reference's OWN CPU path
// (/root/reference/src/plan_manage/include/se3gcopter/{se3gcopter_cpu,trajectory,geoutils,sdlp,quickhull,lbfgs}.hpp),
// compiled where the sources lie, against oracle/eigen_shim (Eigen is not installed here; the shim implements the
// subset of its API those headers use).  No reference source is copied: this file only #includes it and exports C
// entry points.  root_finder.hpp (Sturm sequences via Eigen's eigenvalue/FFT modules) is post-optimisation
// diagnostics only (MinCoPlan_CPU.cpp:131-132) and is replaced by stubs.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <random>
#include <set>
#include <vector>

#include <Eigen/Eigen>

#define ROOT_FINDER_HPP
namespace RootFinder {
inline Eigen::VectorXd polySqr(const Eigen::VectorXd &c) { return c; }
inline double polyVal(const Eigen::VectorXd &, double) { return 1.0; }
inline std::set<double> solvePolynomial(const Eigen::VectorXd &, double, double, double, bool = true) { return {}; }
inline int countRoots(const Eigen::VectorXd &, double, double) { return 0; }
} // namespace RootFinder

// REF_FLAVOUR_GPU (ref_gcopter_gpu_wrap.cpp): the reference's GPU header instead - se3gcopter_gpu.hpp, unmodified, whose `#include <cuda_computer.cuh>` finds
// oracle/frx_dropin/cuda_computer.cuh (the drop-in backed by libfrx.so) - with the same exports under the names refgpu_*.
#define private public          // test access to SE3GCOPTER / MINCO_S3 internals (objectiveFunc, cfgVs, jerkOpt)
#ifdef REF_FLAVOUR_GPU
#include "se3gcopter_gpu.hpp"
#define REFN(name) refgpu_##name
#else
#include "se3gcopter_cpu.hpp"
#define REFN(name) ref_##name
#endif
#undef private
#define REF_EXPORT __attribute__((visibility("default")))

namespace {
struct Cfg {                     // same layout as orc::Config / frx_config
    double rho, total_t, grid_res;
    int qd_intervals, c2_diffeo;
    double horiz_half_len, vert_half_len, safe_margin;
    double vel_max, thr_acc_min, thr_acc_max, body_rate_max, grav_acc;
    double penalty_pvtb[4];
};
Eigen::MatrixXd mat(const double *p, int r, int c) {
    Eigen::MatrixXd m(r, c);
    for (int j = 0; j < c; j++) for (int i = 0; i < r; i++) m(i, j) = p[j * r + i];
    return m;
}
} // namespace

extern "C" {

// setup (se3gcopter_cpu.hpp:1076-1186) on the caller's H-polytopes.  The reference enumerates the V-polytopes itself
// (extractVs -> geoutils::enumerateVs).  With override_vs != 0 the caller's vertex lists replace them afterwards (same
// [v0, v_r - v0] re-basing, se3gcopter_cpu.hpp:1049), so that the hot path can be compared on IDENTICAL inputs: the vertex
// order of enumerateVs depends on an LP with a process-global random permutation.
REF_EXPORT void *REFN(create)(const Cfg *cf, const double *ini, const double *fin, int coarseN, const int *hOff, const double *hRec,
                 const int *vOff, const double *vRec, int override_vs) {
    SE3GCOPTER *g = new SE3GCOPTER();
    std::vector<Eigen::MatrixXd> hPolys;
    for (int i = 0; i < coarseN; i++) hPolys.push_back(mat(hRec + 6 * (size_t)hOff[i], 6, hOff[i + 1] - hOff[i]));
    Eigen::Vector4d w(cf->penalty_pvtb[0], cf->penalty_pvtb[1], cf->penalty_pvtb[2], cf->penalty_pvtb[3]);
    Eigen::MatrixXd iS = mat(ini, 3, 3), fS = mat(fin, 3, 3);
    if (!g->setup(cf->rho, cf->total_t, iS, fS, hPolys, cf->grid_res, cf->qd_intervals, cf->horiz_half_len, cf->vert_half_len,
                  cf->safe_margin, cf->vel_max, cf->thr_acc_min, cf->thr_acc_max, cf->body_rate_max, cf->grav_acc, w,
                  cf->c2_diffeo != 0)) {
        delete g;
        return nullptr;
    }
    if (override_vs) {
        for (int m = 0; m < 2 * coarseN - 1; m++) {
            const int nv = vOff[m + 1] - vOff[m];
            if (nv != (int)g->cfgVs[m].cols()) { delete g; return nullptr; }      // vertex counts must agree (dimFreeP)
            Eigen::MatrixXd V = mat(vRec + 3 * (size_t)vOff[m], 3, nv), ob(3, nv);
            ob << V.col(0), V.rightCols(nv - 1).colwise() - V.col(0);
            g->cfgVs[m] = ob;
        }
    }
    return g;
}
REF_EXPORT void REFN(destroy)(void *h) { delete (SE3GCOPTER *)h; }
REF_EXPORT void REFN(dims)(void *h, int *out4) {
    SE3GCOPTER *g = (SE3GCOPTER *)h;
    out4[0] = g->coarseN; out4[1] = g->fineN; out4[2] = g->dimFreeT; out4[3] = g->dimFreeP;
}
// number of vertices / the vertices (absolute coordinates) of the reference's own V-polytope m (for the f1 row)
REF_EXPORT int REFN(vpoly)(void *h, int m, double *out, int cap) {
    SE3GCOPTER *g = (SE3GCOPTER *)h;
    const Eigen::MatrixXd &V = g->cfgVs[m];
    const int nv = V.cols();
    if (out && nv <= cap)
        for (int j = 0; j < nv; j++) for (int r = 0; r < 3; r++) out[3 * j + r] = V(r, j) + (j > 0 ? V(r, 0) : 0.0);
    return nv;
}
// first half of optimize(): setInitial + backwardT + backwardP (se3gcopter_cpu.hpp:1237-1240)
REF_EXPORT void REFN(initial_guess)(void *h, double *x) {
    SE3GCOPTER *g = (SE3GCOPTER *)h;
    Eigen::Map<Eigen::VectorXd> t(x, g->dimFreeT), p(x + g->dimFreeT, g->dimFreeP);
    g->setInitial(g->cfgVs, g->intervals, g->coarseT, g->innerP);
    SE3GCOPTER::backwardT(g->coarseT, t, g->softT, g->c2dfm);
    SE3GCOPTER::backwardP(g->innerP, g->idxVs, g->cfgVs, p);
}
// the L-BFGS callback itself (se3gcopter_cpu.hpp:961-1000)
REF_EXPORT double REFN(objective)(void *h, const double *x, double *grad) {
    SE3GCOPTER *g = (SE3GCOPTER *)h;
    return SE3GCOPTER::objectiveFunc(g, x, grad, g->dimFreeT + g->dimFreeP);
}
// x -> fine T, coefficients (6N x 3 row-major): forwardT/P + generate (se3gcopter_cpu.hpp:1258-1262)
REF_EXPORT void REFN(forward)(void *h, const double *x, double *T, double *C) {
    SE3GCOPTER *g = (SE3GCOPTER *)h;
    Eigen::Map<const Eigen::VectorXd> t(x, g->dimFreeT), p(x + g->dimFreeT, g->dimFreeP);
    SE3GCOPTER::forwardT(t, g->coarseT, g->softT, g->sumT, g->c2dfm);
    SE3GCOPTER::splitToFineT(g->coarseT, g->intervals, g->fineT);
    SE3GCOPTER::forwardP(p, g->idxVs, g->cfgVs, g->innerP);
    g->jerkOpt.generate(g->innerP, g->fineT);
    const int N = g->fineN;
    for (int i = 0; i < N; i++) T[i] = g->fineT(i);
    for (int r = 0; r < 6 * N; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = g->jerkOpt.b(r, c);
}
// addTimeIntPenalty alone on given (T, C) (se3gcopter_cpu.hpp:188-408), accumulating like cuda_computer::compute
REF_EXPORT void REFN(penalty)(void *h, const double *T, const double *C, double *cost, double *gdT, double *gdC) {
    SE3GCOPTER *g = (SE3GCOPTER *)h;
    const int N = g->fineN;
    g->jerkOpt.T1.resize(N);
    for (int i = 0; i < N; i++) g->jerkOpt.T1(i) = T[i];
    for (int r = 0; r < 6 * N; r++) for (int c = 0; c < 3; c++) g->jerkOpt.b(r, c) = C[r * 3 + c];
    Eigen::VectorXd gT(N);
    Eigen::MatrixXd gC(6 * N, 3);
    for (int i = 0; i < N; i++) gT(i) = gdT[i];
    for (int r = 0; r < 6 * N; r++) for (int c = 0; c < 3; c++) gC(r, c) = gdC[r * 3 + c];
    double cst = *cost;
    g->jerkOpt.addTimeIntPenalty(g->cons, g->idxHs, g->cfgHs, g->ellipsoid, g->safeMargin, g->vMax, g->thrAccMin, g->thrAccMax,
                                 g->bdrMax, g->gAcc, g->chi, cst, gT, gC);
    *cost = cst;
    for (int i = 0; i < N; i++) gdT[i] = gT(i);
    for (int r = 0; r < 6 * N; r++) for (int c = 0; c < 3; c++) gdC[r * 3 + c] = gC(r, c);
}
// the whole SE3GCOPTER::optimize (se3gcopter_cpu.hpp:1230-1268); returns the jerk cost like the reference
REF_EXPORT double REFN(optimize)(void *h, double relCostTol, double *C, double *T) {
    SE3GCOPTER *g = (SE3GCOPTER *)h;
    Trajectory traj;
    const double jc = g->optimize(traj, relCostTol);
    const int N = g->fineN;
    for (int i = 0; i < N; i++) T[i] = g->fineT(i);
    for (int r = 0; r < 6 * N; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = g->jerkOpt.b(r, c);
    return jc;
}
#ifdef REF_FLAVOUR_GPU
// calls of cuda_computer::compute served so far (the reference's GPU header makes TWO per evaluation, se3gcopter_gpu.hpp:219-227), and kill_kernel (:907-909)
REF_EXPORT long refgpu_compute_calls(void *h) { return ((SE3GCOPTER *)h)->jerkOpt.paraller.n_compute; }
REF_EXPORT void refgpu_kill_kernel(void *h) { ((SE3GCOPTER *)h)->kill_kernel(); }
#endif
}
