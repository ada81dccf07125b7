// Host side of SE3GCOPTER::setup and of the first half of SE3GCOPTER::optimize (setInitial / backwardT / backwardP): what the library keeps per
// candidate and the reference's initial guess.  Host code only (frx_api.cpp; tests/hostcheck builds it alone to check it on a CPU-only box).
//
// The reference runs this work serially inside its plan timer (MinCoPlan_CPU.cpp:113-126): per candidate N - 1 nested L-BFGS solves, one per
// waypoint (backwardP, CPU.hpp:777-813), each independent of the others.  Here the solves of the whole batch are ONE flat task list over
// (candidate, waypoint) for the thread pool; every solve runs the same arithmetic in the same order as before, so x0 does not depend on the
// number of threads.
#pragma once
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/frx.h"
#include "frx_host_pool.hpp"
#include "frx_lbfgs.hpp"

namespace frx {

// ---- per-candidate host description (what setup() keeps for the initial guess) ----
struct HostCand {
    int coarseN = 0, fineN = 0, dimT = 0, dimP = 0;
    double iState[9], fState[9];                 // clipped copies (CPU.hpp:1166-1170)
    std::vector<std::vector<double>> cfgVs;      // [v0, v_r - v0] (CPU.hpp:1049)
    std::vector<int> intervals, idxVs;
};

inline void poly_centre(const std::vector<double> &V, double *c) {     // CPU.hpp:1018-1019 / 1206-1207
    const int k = (int)(V.size() / 3) - 1;
    for (int r = 0; r < 3; r++) {
        double s = 0.0;
        for (int a = 0; a < k; a++) s += V[3 * (a + 1) + r];
        c[r] = s / (1.0 + k) + V[r];
    }
}
inline double dist3(const double *a, const double *b) {
    return std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
}

// The host part of setup() for one candidate: V-polytopes as [v0, v_r - v0], gridMesh (CPU.hpp:1003-1029), the index map of the
// waypoints (CPU.hpp:1129-1152) and the legal boundary speed on the copies (CPU.hpp:1166-1170).  v_off points at this candidate's
// first polytope (2 cN - 1 of them).  Returns FRX_OK or the error of a polytope without vertices.
inline int host_cand_init(HostCand &hc, const frx_config &cfg, bool softT, int cN, const double *ini9, const double *fin9, const int *v_off, const double *v_rec) {
    hc.coarseN = cN;
    std::memcpy(hc.iState, ini9, sizeof(hc.iState));
    std::memcpy(hc.fState, fin9, sizeof(hc.fState));
    hc.cfgVs.resize(2 * cN - 1);
    for (int m = 0; m < 2 * cN - 1; m++) {
        const int beg = v_off[m], nv = v_off[m + 1] - beg;
        if (nv < 1) return FRX_ERR_EMPTY_POLYTOPE;
        hc.cfgVs[m].resize(3 * (size_t)nv);
        const double *v = v_rec + 3 * (size_t)beg;
        for (int r = 0; r < 3; r++) hc.cfgVs[m][r] = v[r];
        for (int a = 1; a < nv; a++)
            for (int r = 0; r < 3; r++) hc.cfgVs[m][3 * a + r] = v[3 * a + r] - v[r];
    }
    hc.intervals.assign(cN, 1);
    {
        double lastP[3], curP[3] = {hc.iState[0], hc.iState[1], hc.iState[2]};
        for (int i = 0; i < cN; i++) {
            std::memcpy(lastP, curP, sizeof(curP));
            if (i < cN - 1) poly_centre(hc.cfgVs[2 * i + 1], curP);
            else { curP[0] = hc.fState[0]; curP[1] = hc.fState[1]; curP[2] = hc.fState[2]; }
            const int cur = (int)std::ceil(dist3(curP, lastP) / cfg.grid_res);
            hc.intervals[i] = cur > 0 ? cur : 1;
        }
    }
    hc.fineN = 0;
    for (int i = 0; i < cN; i++) hc.fineN += hc.intervals[i];
    hc.dimT = softT ? cN : cN - 1;                                       // CPU.hpp:1131
    hc.idxVs.assign(std::max(hc.fineN - 1, 0), 0);
    hc.dimP = 0;
    int offset = 0;
    for (int i = 0; i < cN; i++)
        for (int j = 0; j < hc.intervals[i]; j++) {
            int vm = -1;
            if (j < hc.intervals[i] - 1) vm = 2 * i;
            else if (i < cN - 1) vm = 2 * i + 1;
            if (vm >= 0) { hc.idxVs[offset] = vm; hc.dimP += (int)(hc.cfgVs[vm].size() / 3) - 1; }
            offset++;
        }
    for (double *st : {hc.iState, hc.fState}) {
        const double tn = std::sqrt(st[3] * st[3] + st[4] * st[4] + st[5] * st[5]);
        const double sc = tn > cfg.vel_max ? (cfg.vel_max / tn) : 1.0;
        for (int r = 0; r < 3; r++) st[3 + r] *= sc;
    }
    return FRX_OK;
}

// objectiveNLS (CPU.hpp:749-774): squared distance between a target point and the image of the
// sphere parameterisation of one V-polytope; pobs = [target, v0, edges...]
inline double nls_objective(const double *pobs, const double *x, double *grad, int n, double *r, double *gdr) {
    double qn = 0.0;
    for (int a = 0; a < n; a++) qn += x[a] * x[a];
    const double qp1 = qn + 1.0, qp1sq = qp1 * qp1, sc = 2.0 / qp1;
    for (int a = 0; a < n; a++) r[a] = sc * x[a];
    double delta[3];
    for (int q = 0; q < 3; q++) {
        double s = 0.0;
        for (int a = 0; a < n; a++) s += pobs[3 * (a + 2) + q] * (r[a] * r[a]);
        delta[q] = s + pobs[3 + q] - pobs[q];
    }
    const double cost = delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2];
    const double g3[3] = {2 * delta[0], 2 * delta[1], 2 * delta[2]};
    for (int a = 0; a < n; a++)
        gdr[a] = (pobs[3 * (a + 2)] * g3[0] + pobs[3 * (a + 2) + 1] * g3[1] + pobs[3 * (a + 2) + 2] * g3[2]) * r[a] * 2.0;
    double gq = 0.0;
    for (int a = 0; a < n; a++) gq += gdr[a] * x[a];
    for (int a = 0; a < n; a++) grad[a] = gdr[a] * 2.0 / qp1 - x[a] * 4.0 * gq / qp1sq;
    return cost;
}

// setInitial + backwardT for one candidate (CPU.hpp:1188-1228, 679-726): the tau block of x and the interior waypoints inP (3 per waypoint)
inline void initial_guess_times_and_points(const frx_config &cfg, bool softT, const HostCand &hc, double *x, double *inP) {
    const double vAlloc = std::min(cfg.vel_max, 10.0);                   // maxSpeedForAllocatiion, CPU.hpp:1193
    const int M = hc.coarseN;
    std::vector<double> vecT(M);
    double lastP[3], curP[3] = {hc.iState[0], hc.iState[1], hc.iState[2]}, delta[3];
    int offset = 0;
    for (int i = 0; i < M; i++) {
        std::memcpy(lastP, curP, sizeof(curP));
        const int interv = hc.intervals[i];
        if (i < M - 1) poly_centre(hc.cfgVs[2 * i + 1], curP);
        else { curP[0] = hc.fState[0]; curP[1] = hc.fState[1]; curP[2] = hc.fState[2]; }
        for (int r = 0; r < 3; r++) delta[r] = curP[r] - lastP[r];
        vecT[i] = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]) / vAlloc;
        for (int r = 0; r < 3; r++) delta[r] /= interv;
        const int cnt = (i < M - 1) ? interv : interv - 1;
        for (int j = 0; j < cnt; j++) {
            for (int r = 0; r < 3; r++) inP[offset * 3 + r] = (j + 1) * delta[r] + lastP[r];
            offset++;
        }
    }
    const bool c2 = cfg.c2_diffeo != 0;
    if (softT) {
        for (int i = 0; i < M; i++)
            x[i] = c2 ? (vecT[i] > 1.0 ? (std::sqrt(2.0 * vecT[i] - 1.0) - 1.0) : (1.0 - std::sqrt(2.0 / vecT[i] - 1.0)))
                      : std::log(vecT[i]);
    } else {
        for (int i = 0; i < M - 1; i++) {
            const double r = vecT[i] / vecT[M - 1];
            x[i] = c2 ? (r > 1.0 ? (std::sqrt(2.0 * r - 1.0) - 1.0) : (1.0 - std::sqrt(2.0 / r - 1.0))) : std::log(r);
        }
    }
}

// backwardP for ONE waypoint (CPU.hpp:777-813): a tiny L-BFGS (default parameters, g_epsilon = FLT_EPSILON, 128 iterations) from
// 1 / (sqrt(k + 1) + 1) on the sphere parameterisation of the waypoint's polytope V = [v0, edges]; xi = the waypoint's k variables.
struct NlsScratch { std::vector<double> pobs, grad, r, gdr; Solver s; };
inline void initial_guess_waypoint(const std::vector<double> &V, const double *target, double *xi, NlsScratch &w) {
    frx_lbfgs_params nls;
    lbfgs_defaults(nls);
    nls.g_epsilon = FLT_EPSILON;
    nls.max_iterations = 128;
    const int k = (int)(V.size() / 3) - 1;
    for (int a = 0; a < k; a++) xi[a] = 1.0 / (std::sqrt(k + 1.0) + 1.0);
    w.pobs.resize(3 * (size_t)(k + 2)); w.r.resize(k); w.gdr.resize(k);
    for (int q = 0; q < 3; q++) w.pobs[q] = target[q];
    std::memcpy(&w.pobs[3], V.data(), sizeof(double) * 3 * (k + 1));
    w.grad.assign(k, 0.0);
    w.s.start(k, xi, w.grad.data(), nls);
    while (!w.s.done()) w.s.feed(nls_objective(w.pobs.data(), xi, w.grad.data(), k, w.r.data(), w.gdr.data()));
}

// The reference's initial guess for a whole batch: x0 [xoff[B]].  Tasks = every (candidate, waypoint) of the batch.
inline void initial_guess_batch(const frx_config &cfg, bool softT, const std::vector<HostCand> &cand, const int *xoff, double *x0) {
    const int B = (int)cand.size();
    std::vector<int> wbase(B + 1, 0);
    for (int b = 0; b < B; b++) wbase[b + 1] = wbase[b] + std::max(cand[b].fineN - 1, 0);
    const int W = wbase[B];
    std::vector<double> inP(3 * (size_t)std::max(W, 1));
    std::vector<int> wcand(W), wxi(W);                                   // per waypoint: its candidate, the offset of its variables in x0
    for (int b = 0; b < B; b++) {
        const HostCand &hc = cand[b];
        initial_guess_times_and_points(cfg, softT, hc, x0 + xoff[b], inP.data() + 3 * (size_t)wbase[b]);
        int j = xoff[b] + hc.dimT;
        for (int i = 0; i < hc.fineN - 1; i++) {
            wcand[wbase[b] + i] = b; wxi[wbase[b] + i] = j;
            j += (int)(hc.cfgVs[hc.idxVs[i]].size() / 3) - 1;
        }
    }
    // (a solve takes ~0.4 us per waypoint variable: the whole headline batch - 2016 solves - is 0.8 ms on one core; threads only for large batches)
    const int nt = setup_threads(W, 0.4);
    const auto tdbg0 = std::chrono::steady_clock::now();
    std::vector<NlsScratch> scratch(nt);
    const int CH = 16;                                                    // waypoints per task: neighbouring parts of x0
    TaskPool::get().run((W + CH - 1) / CH, nt, [&](int task, int worker) {
        for (int w = task * CH, hi = std::min(W, w + CH); w < hi; w++) {
            const HostCand &hc = cand[wcand[w]];
            initial_guess_waypoint(hc.cfgVs[hc.idxVs[w - wbase[wcand[w]]]], inP.data() + 3 * (size_t)w, x0 + wxi[w], scratch[worker]);
        }
    });
    if (std::getenv("FRX_SETUP_TIMING")) fprintf(stderr, "[frx setup] initial guess: %d waypoint solves on %d threads, %.3f ms\n", W, nt, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tdbg0).count());
}

} // namespace frx
