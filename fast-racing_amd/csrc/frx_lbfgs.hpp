// Host L-BFGS as RESUMABLE state machines.
//
// The reference's solver is a blocking call with an evaluation callback
// (lbfgs::lbfgs_optimize + lbfgs_evaluate_t, lbfgs.hpp:1103, :223), which cannot batch: B
// candidates would need B separate device round trips per step.  Here every candidate owns
// one solver object; the driver collects the trial points of all candidates that are waiting
// for an objective value, evaluates them in ONE batched device pass, and feeds the values back
// (SURVEY.md §7.1-6).
//
//   LineSearch  the scalar logic of line_search_morethuente (lbfgs.hpp:730-938, with
//               update_trial_interval :520-728) and line_search_backtracking (:940-1033), split at the
//               evaluation so it can be resumed.  It never touches a vector: it consumes
//               (f, dg = g.d) of a trial step and produces the next step or a verdict.
//   Solver      host vectors: x, g, d, the (s, y) history and the two-loop recursion live in host
//               memory and run in exactly the reference's operation order, so its iterates are
//               bit-identical to lbfgs_optimize (tests/test_lbfgs.py checks that against the
//               reference header itself).
//   SolverDV    device vectors: the SAME control flow, but every O(n) operation is a command to
//               the device (frx_lbfgs_kernels.hpp) and only scalars cross PCIe: per round and candidate
//               the host sends {flags, step, history slot} and receives {f, g.d, x.x, g.g, g.d_new}.
//               The decisions (Moré–Thuente / backtracking, convergence and stop tests, error codes)
//               stay on the host, as in the reference; dot products are summed in a different
//               (fixed, parallel) order, so iterates agree with Solver to rounding, not bitwise.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/frx.h"

// The scalar line-search logic below also runs ON THE DEVICE (the leader of the resident round kernel predicts the host's next command,
// frx_round_kernel.hpp): same source for both compilers, and no floating-point contraction in either - the host side and the CPU oracle are
// built with -ffp-contract=off (like the reference, whose own build has no FMA: -O3 without -march, CMakeLists.txt:7), hipcc is told per function.
#if defined(__HIPCC__)
#define FRX_LS_HD __host__ __device__ __forceinline__
#define FRX_LS_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define FRX_LS_HD inline
#define FRX_LS_NO_CONTRACT
#endif

namespace frx {

enum {   // numeric values of lbfgs.hpp:149-206
    LB_CONVERGENCE = 0, LB_STOP = 1, LB_ALREADY_MINIMIZED = 2,
    LBERR_UNKNOWN = -1024, LBERR_LOGIC, LBERR_CANCELED, LBERR_INVALID_N, LBERR_INVALID_MEMSIZE,
    LBERR_INVALID_GEPSILON, LBERR_INVALID_TESTPERIOD, LBERR_INVALID_DELTA, LBERR_INVALID_MINSTEP,
    LBERR_INVALID_MAXSTEP, LBERR_INVALID_FDECCOEFF, LBERR_INVALID_SCURVCOEFF, LBERR_INVALID_XTOL,
    LBERR_INVALID_MAXLINESEARCH, LBERR_OUTOFINTERVAL, LBERR_INCORRECT_TMINMAX, LBERR_ROUNDING,
    LBERR_MINIMUMSTEP, LBERR_MAXIMUMSTEP, LBERR_MAXIMUMLINESEARCH, LBERR_MAXIMUMITERATION,
    LBERR_WIDTHTOOSMALL, LBERR_INVALIDPARAMETERS, LBERR_INCREASEGRADIENT
};

inline void lbfgs_defaults(frx_lbfgs_params &p) {      // _default_param, lbfgs.hpp:128-140
    p.mem_size = 8; p.g_epsilon = 1e-5; p.past = 0; p.delta = 1e-5; p.max_iterations = 0; p.max_linesearch = 40;
    p.min_step = 1e-20; p.max_step = 1e20; p.f_dec_coeff = 1e-4; p.s_curv_coeff = 0.9; p.xtol = 1e-16;
}
inline int lbfgs_check(int n, const frx_lbfgs_params &p) {   // lbfgs.hpp:1143-1186
    if (n <= 0) return LBERR_INVALID_N;
    if (p.mem_size <= 0) return LBERR_INVALID_MEMSIZE;
    if (p.g_epsilon < 0.) return LBERR_INVALID_GEPSILON;
    if (p.past < 0) return LBERR_INVALID_TESTPERIOD;
    if (p.delta < 0.) return LBERR_INVALID_DELTA;
    if (p.min_step < 0.) return LBERR_INVALID_MINSTEP;
    if (p.max_step < p.min_step) return LBERR_INVALID_MAXSTEP;
    if (p.f_dec_coeff < 0.) return LBERR_INVALID_FDECCOEFF;
    if (p.s_curv_coeff <= p.f_dec_coeff || 1. <= p.s_curv_coeff) return LBERR_INVALID_SCURVCOEFF;
    if (p.xtol < 0.) return LBERR_INVALID_XTOL;
    if (p.max_linesearch <= 0) return LBERR_INVALID_MAXLINESEARCH;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Scalar line-search logic (no vectors).  Protocol:
//    r = mt_begin(step, f0, dginit)       r != 0: error code, no evaluation wanted
//    loop:  evaluate f, dg at  x = xp + step()*d ;  r = mt_feed(f, dg)
//           r == PENDING: another trial at step();  otherwise r = ls (>0 evaluations, <0 error)
//    same with bt_begin / bt_feed.
// ---------------------------------------------------------------------------------------------
// 24-bit fold of a step's bit pattern: rides in the slot / pair-count fields of a resident-kernel command that carries TRIAL without ADVANCE
// (those fields are unused there), so that the leader can confirm a predicted trial step from the command WORD alone (frx_round_kernel.hpp)
FRX_LS_HD unsigned dv_step_hash(double step) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, step);
    return (unsigned)((b ^ (b >> 24) ^ (b >> 48)) & 0xFFFFFFull);
}

class LineSearch {
public:
    static constexpr int PENDING = 1 << 30;
    FRX_LS_HD double step() const { return stp; }

    FRX_LS_HD int mt_begin(const frx_lbfgs_params &pm, double step0, double f0, double dginit0) { FRX_LS_NO_CONTRACT     // lbfgs.hpp:743-788
        count = 0; brackt = 0; stage1 = 1; uinfo = 0;
        stp = step0;
        if (stp <= 0.) return LBERR_INVALIDPARAMETERS;
        dginit = dginit0;
        if (0 < dginit) return LBERR_INCREASEGRADIENT;
        finit = f0;
        dgtest = pm.f_dec_coeff * dginit;
        width = pm.max_step - pm.min_step;
        prev_width = 2.0 * width;
        stx = sty = 0.;
        fxl = fy = finit;
        dgx = dgy = dginit;
        mt_propose(pm);
        return 0;
    }
    FRX_LS_HD int mt_feed(const frx_lbfgs_params &pm, double f, double dg) { FRX_LS_NO_CONTRACT                          // lbfgs.hpp:829-935
        const double ftest1 = finit + stp * dgtest;
        ++count;
        if ((std::isinf(f) || std::isnan(f)) || (brackt && ((stp <= stmin || stmax <= stp) || uinfo != 0))) return LBERR_ROUNDING;
        if (stp == pm.max_step && f <= ftest1 && dg <= dgtest) return LBERR_MAXIMUMSTEP;
        if (stp == pm.min_step && (ftest1 < f || dgtest <= dg)) return LBERR_MINIMUMSTEP;
        if (brackt && (stmax - stmin) <= pm.xtol * stmax) return LBERR_WIDTHTOOSMALL;
        if (pm.max_linesearch <= count) return LBERR_MAXIMUMLINESEARCH;
        if (f <= ftest1 && std::fabs(dg) <= pm.s_curv_coeff * (-dginit)) return count;
        if (stage1 && f <= ftest1 && (pm.f_dec_coeff <= pm.s_curv_coeff ? pm.f_dec_coeff : pm.s_curv_coeff) * dginit <= dg)
            stage1 = 0;
        // (one call of update_trial on LOCAL copies for both branches: with the members passed by reference from two call sites the device
        // compiler selected between member addresses and put the whole object into scratch memory)
        const bool modified = stage1 && ftest1 < f && f <= fxl;            // the modified function psi of lbfgs.hpp:870-905
        double xs = stx, ys = sty, tt = stp, fxs, dxs, fys, dys, ft, dt;
        int br = brackt;
        if (modified) { ft = f - stp * dgtest; fxs = fxl - stx * dgtest; fys = fy - sty * dgtest; dt = dg - dgtest; dxs = dgx - dgtest; dys = dgy - dgtest; }
        else { ft = f; fxs = fxl; fys = fy; dt = dg; dxs = dgx; dys = dgy; }
        uinfo = update_trial(xs, fxs, dxs, ys, fys, dys, tt, ft, dt, stmin, stmax, br);
        stx = xs; sty = ys; stp = tt; brackt = br;
        if (modified) { fxl = fxs + stx * dgtest; fy = fys + sty * dgtest; dgx = dxs + dgtest; dgy = dys + dgtest; }
        else { fxl = fxs; fy = fys; dgx = dxs; dgy = dys; }
        if (brackt) {
            if (0.66 * prev_width <= std::fabs(sty - stx)) stp = stx + 0.5 * (sty - stx);
            prev_width = width;
            width = std::fabs(sty - stx);
        }
        mt_propose(pm);
        return PENDING;
    }

    // What mt_begin followed by ONE mt_feed decides for the first trial of a search when that trial is ACCEPTED, on their operands in their
    // order: true exactly when mt_begin(pm, step0, f0, dginit0) == 0 and the mt_feed(pm, f, dg) after it returns 1 (= count) - for a step
    // strictly between min_step and max_step (the caller's precondition; the tests on those two and on a bracket cannot fire then).  The
    // leader of the resident round kernel asks this first: four rounds in five end this way and the state machine need not run at all
    // (tests/test_lbfgs.py checks the equivalence on the host).
    FRX_LS_HD static bool first_trial_accepted(const frx_lbfgs_params &pm, double step0, double f0, double dginit0, double f, double dg) { FRX_LS_NO_CONTRACT
        if (!(step0 > 0.) || 0 < dginit0) return false;                    // mt_begin's two refusals
        const double dgtest = pm.f_dec_coeff * dginit0;
        const double ftest1 = f0 + step0 * dgtest;
        if (std::isinf(f) || std::isnan(f)) return false;
        if (pm.max_linesearch <= 1) return false;
        return f <= ftest1 && std::fabs(dg) <= pm.s_curv_coeff * (-dginit0);
    }

    FRX_LS_HD int bt_begin(double step0, double f0, double dginit0, const frx_lbfgs_params &pm) { FRX_LS_NO_CONTRACT      // lbfgs.hpp:954-974
        count = 0;
        stp = step0;
        if (stp <= 0.) return LBERR_INVALIDPARAMETERS;
        dginit = dginit0;
        if (0 < dginit) return LBERR_INCREASEGRADIENT;
        finit = f0;
        dgtest = pm.f_dec_coeff * dginit;
        return 0;
    }
    // dg is only consulted when the sufficient-decrease test passes, like the reference (lbfgs.hpp:986-1010)
    FRX_LS_HD int bt_feed(const frx_lbfgs_params &pm, double f, double dg) { FRX_LS_NO_CONTRACT
        const double dec = 0.5, inc = 2.1;
        double wd;
        ++count;
        if (f > finit + stp * dgtest) wd = dec;
        else {
            if (dg < pm.s_curv_coeff * dginit) wd = inc;
            else if (dg > -pm.s_curv_coeff * dginit) wd = dec;
            else return count;
        }
        if (stp < pm.min_step) return LBERR_MINIMUMSTEP;
        if (stp > pm.max_step) return LBERR_MAXIMUMSTEP;
        if (pm.max_linesearch <= count) return LBERR_MAXIMUMLINESEARCH;
        stp *= wd;
        return PENDING;
    }
    FRX_LS_HD bool bt_needs_dg(double f) const { FRX_LS_NO_CONTRACT return !(f > finit + stp * dgtest); }

private:
    int count = 0, brackt = 0, stage1 = 0, uinfo = 0;
    double stp = 0, stx = 0, fxl = 0, dgx = 0, sty = 0, fy = 0, dgy = 0, finit = 0, dginit = 0, dgtest = 0, width = 0, prev_width = 0,
           stmin = 0, stmax = 0;

    FRX_LS_HD void mt_propose(const frx_lbfgs_params &pm) { FRX_LS_NO_CONTRACT             // loop head up to the evaluation, lbfgs.hpp:790-826
        if (brackt) { stmin = stx <= sty ? stx : sty; stmax = stx >= sty ? stx : sty; }
        else        { stmin = stx; stmax = stp + 4.0 * (stp - stx); }
        if (stp < pm.min_step) stp = pm.min_step;
        if (pm.max_step < stp) stp = pm.max_step;
        if ((brackt && ((stp <= stmin || stmax <= stp) || pm.max_linesearch <= count + 1 || uinfo != 0)) ||
            (brackt && (stmax - stmin <= pm.xtol * stmax)))
            stp = stx;
    }

    // ---- interpolants, lbfgs.hpp:318-406 ----
    FRX_LS_HD static double cubic_min(double u, double fu, double du, double v, double fv, double dv) { FRX_LS_NO_CONTRACT
        const double dd = v - u;
        const double theta = (fu - fv) * 3 / dd + du + dv;
        double p = std::fabs(theta), q = std::fabs(du), r = std::fabs(dv);
        double s = p >= q ? p : q;
        s = s >= r ? s : r;
        const double a = theta / s;
        double gamm = s * std::sqrt(a * a - (du / s) * (dv / s));
        if (v < u) gamm = -gamm;
        p = gamm - du + theta;
        q = gamm - du + gamm + dv;
        r = p / q;
        return u + r * dd;
    }
    FRX_LS_HD static double cubic_min_bounded(double u, double fu, double du, double v, double fv, double dv, double xmin, double xmax) { FRX_LS_NO_CONTRACT
        const double dd = v - u;
        const double theta = (fu - fv) * 3 / dd + du + dv;
        double p = std::fabs(theta), q = std::fabs(du), r = std::fabs(dv);
        double s = p >= q ? p : q;
        s = s >= r ? s : r;
        const double a = theta / s;
        double gamm = a * a - (du / s) * (dv / s);
        gamm = gamm > 0 ? s * std::sqrt(gamm) : 0;
        if (u < v) gamm = -gamm;
        p = gamm - dv + theta;
        q = gamm - dv + gamm + du;
        r = p / q;
        if (r < 0. && gamm != 0.) return v - r * dd;
        return a < 0 ? xmax : xmin;
    }
    FRX_LS_HD static double quad_min(double u, double fu, double du, double v, double fv) { FRX_LS_NO_CONTRACT
        const double a = v - u;
        return u + du / ((fu - fv) / a + du) / 2 * a;
    }
    FRX_LS_HD static double secant_min(double u, double du, double v, double dv) { FRX_LS_NO_CONTRACT
        const double a = u - v;
        return v + dv / (dv - du) * a;
    }
    // update_trial_interval, lbfgs.hpp:520-728
    FRX_LS_HD static int update_trial(double &xs, double &fxs, double &dxs, double &ys, double &fys, double &dys,
                            double &t, double &ft, double &dt, double tmin, double tmax, int &br) { FRX_LS_NO_CONTRACT
        int bound;
        const int dsign = dt * (dxs / std::fabs(dxs)) < 0.;
        double mc, mq, newt;
        if (br) {
            if (t <= (xs <= ys ? xs : ys) || (xs >= ys ? xs : ys) <= t) return LBERR_OUTOFINTERVAL;
            if (0. <= dxs * (t - xs)) return LBERR_INCREASEGRADIENT;
            if (tmax < tmin) return LBERR_INCORRECT_TMINMAX;
        }
        if (fxs < ft) {
            br = 1; bound = 1;
            mc = cubic_min(xs, fxs, dxs, t, ft, dt);
            mq = quad_min(xs, fxs, dxs, t, ft);
            newt = (std::fabs(mc - xs) < std::fabs(mq - xs)) ? mc : mc + 0.5 * (mq - mc);
        } else if (dsign) {
            br = 1; bound = 0;
            mc = cubic_min(xs, fxs, dxs, t, ft, dt);
            mq = secant_min(xs, dxs, t, dt);
            newt = (std::fabs(mc - t) > std::fabs(mq - t)) ? mc : mq;
        } else if (std::fabs(dt) < std::fabs(dxs)) {
            bound = 1;
            mc = cubic_min_bounded(xs, fxs, dxs, t, ft, dt, tmin, tmax);
            mq = secant_min(xs, dxs, t, dt);
            if (br) newt = (std::fabs(t - mc) < std::fabs(t - mq)) ? mc : mq;
            else    newt = (std::fabs(t - mc) > std::fabs(t - mq)) ? mc : mq;
        } else {
            bound = 0;
            if (br)          newt = cubic_min(t, ft, dt, ys, fys, dys);
            else if (xs < t) newt = tmax;
            else             newt = tmin;
        }
        {   // interval update (lbfgs.hpp:690-712) as selections: the branchy form made the device compiler index the six values in scratch memory
            const bool up = fxs < ft, sw = !up && dsign;
            const double nys = up ? t : (sw ? xs : ys), nfys = up ? ft : (sw ? fxs : fys), ndys = up ? dt : (sw ? dxs : dys);
            const double nxs = up ? xs : t, nfxs = up ? fxs : ft, ndxs = up ? dxs : dt;
            xs = nxs; fxs = nfxs; dxs = ndxs; ys = nys; fys = nfys; dys = ndys;
        }
        if (tmax < newt) newt = tmax;
        if (newt < tmin) newt = tmin;
        if (br && bound) {
            mq = xs + 0.66 * (ys - xs);
            if (xs < ys) { if (mq < newt) newt = mq; }
            else         { if (newt < mq) newt = mq; }
        }
        t = newt;
        return 0;
    }
};

// ---------------------------------------------------------------------------------------------
// Host-vector solver: bit-identical to lbfgs::lbfgs_optimize.
// ---------------------------------------------------------------------------------------------
class Solver {
public:
    // x and g are caller-owned storage of n doubles each: the solver writes the next trial point into x and
    // expects the gradient at that point in g.
    void start(int n_, double *x_, double *g_, const frx_lbfgs_params &pm_) {
        n = n_; x = x_; g = g_; pm = pm_;
        m = pm.mem_size;
        ret = lbfgs_check(n, pm);
        evals = 0; k = 0; fx = 0.0;
        if (ret != 0) { phase = DONE; return; }
        xp.assign(n, 0.0); gp.assign(n, 0.0); d.assign(n, 0.0);
        S.assign((size_t)m * n, 0.0); Y.assign((size_t)m * n, 0.0);
        alpha.assign(m, 0.0); ysv.assign(m, 0.0);
        pf.assign(pm.past > 0 ? pm.past : 0, 0.0);
        phase = WAIT_INITIAL;                 // first evaluation at the start point (lbfgs.hpp:1211)
    }
    bool done() const { return phase == DONE; }
    int status() const { return ret; }
    int iterations() const { return k; }
    int evaluations() const { return evals; }
    double value() const { return fx; }

    // the objective value at the current x (gradient already in g)
    void feed(double f) {
        ++evals;
        switch (phase) {
        case WAIT_INITIAL: after_initial(f); break;
        case WAIT_MT: {
            fx = f;
            const double dg = vdot(g, d.data(), n);                      // lbfgs.hpp:830
            const int r = ls.mt_feed(pm, f, dg);
            if (r == LineSearch::PENDING) propose(); else linesearch_result(r, true);
            break;
        }
        case WAIT_BT: {
            fx = f;
            const double dg = ls.bt_needs_dg(f) ? vdot(g, d.data(), n) : 0.0;   // lbfgs.hpp:993
            const int r = ls.bt_feed(pm, f, dg);
            if (r == LineSearch::PENDING) propose(); else linesearch_result(r, false);
            break;
        }
        default: break;
        }
    }

private:
    enum Phase { WAIT_INITIAL, WAIT_MT, WAIT_BT, DONE };
    int n = 0, m = 0;
    double *x = nullptr, *g = nullptr;
    frx_lbfgs_params pm;
    Phase phase = DONE;
    int ret = 0, k = 0, end = 0, evals = 0;
    double fx = 0, step = 0, stepp = 0, fp = 0;
    std::vector<double> xp, gp, d, S, Y, alpha, ysv, pf;
    LineSearch ls;

    static double vdot(const double *a, const double *b, int n) {
        double s = 0.;
        for (int i = 0; i < n; ++i) s += a[i] * b[i];
        return s;
    }
    static void vadd(double *y, const double *v, double c, int n) {
        for (int i = 0; i < n; ++i) y[i] += c * v[i];
    }
    void finish(int code) { ret = code; phase = DONE; }
    void propose() {                                   // x <- xp + stp * d   (lbfgs.hpp:825-826, 977-978)
        std::memcpy(x, xp.data(), sizeof(double) * n);
        vadd(x, d.data(), ls.step(), n);
    }

    // lbfgs.hpp:1211-1246
    void after_initial(double f) {
        fx = f;
        if (!pf.empty()) pf[0] = fx;
        for (int i = 0; i < n; ++i) d[i] = -g[i];
        double xnorm = std::sqrt(vdot(x, x, n)), gnorm = std::sqrt(vdot(g, g, n));
        if (xnorm < 1.0) xnorm = 1.0;
        if (gnorm / xnorm <= pm.g_epsilon) { finish(LB_ALREADY_MINIMIZED); return; }
        step = 1.0 / std::sqrt(vdot(d.data(), d.data(), n));
        k = 1;
        end = 0;
        begin_iteration();
    }
    // lbfgs.hpp:1248-1268
    void begin_iteration() {
        std::memcpy(xp.data(), x, sizeof(double) * n);
        std::memcpy(gp.data(), g, sizeof(double) * n);
        stepp = step;
        fp = fx;
        const int r = ls.mt_begin(pm, step, fx, vdot(gp.data(), d.data(), n));
        if (r != 0) { linesearch_result(r, true); return; }
        phase = WAIT_MT;
        propose();
    }
    // lbfgs.hpp:1270-1293
    void linesearch_result(int lsr, bool from_mt) {
        step = ls.step();
        if (lsr < 0 && from_mt) {
            step = stepp;
            fx = fp;
            const int r = ls.bt_begin(step, fx, vdot(gp.data(), d.data(), n), pm);
            if (r != 0) { linesearch_result(r, false); return; }
            phase = WAIT_BT;
            propose();
            return;
        }
        if (lsr < 0) {
            std::memcpy(x, xp.data(), sizeof(double) * n);
            std::memcpy(g, gp.data(), sizeof(double) * n);
            // value() must belong to the point that is returned: the reference leaves fx of the last REJECTED trial in *ptr_fx
            // (lbfgs.hpp:1287-1291, 1429) and its only caller discards it (CPU.hpp:1249); here the value ranks candidates
            fx = fp;
            finish(lsr);
            return;
        }
        after_linesearch();
    }
    // lbfgs.hpp:1295-1419
    void after_linesearch() {
        double xnorm = std::sqrt(vdot(x, x, n)), gnorm = std::sqrt(vdot(g, g, n));
        if (xnorm < 1.0) xnorm = 1.0;
        if (gnorm / xnorm <= pm.g_epsilon) { finish(LB_CONVERGENCE); return; }
        if (!pf.empty()) {
            if (pm.past <= k) {
                const double rate = (pf[k % pm.past] - fx) / fx;
                if (std::fabs(rate) < pm.delta) { finish(LB_STOP); return; }
            }
            pf[k % pm.past] = fx;
        }
        if (pm.max_iterations != 0 && pm.max_iterations < k + 1) { finish(LBERR_MAXIMUMITERATION); return; }

        double *se = &S[(size_t)end * n], *ye = &Y[(size_t)end * n];
        for (int i = 0; i < n; ++i) se[i] = x[i] - xp[i];
        for (int i = 0; i < n; ++i) ye[i] = g[i] - gp[i];
        const double ys = vdot(ye, se, n), yy = vdot(ye, ye, n);
        ysv[end] = ys;
        const int bound = (m <= k) ? m : k;
        ++k;
        end = (end + 1) % m;
        for (int i = 0; i < n; ++i) d[i] = -g[i];
        int j = end;
        for (int i = 0; i < bound; ++i) {
            j = (j + m - 1) % m;
            alpha[j] = vdot(&S[(size_t)j * n], d.data(), n);
            alpha[j] /= ysv[j];
            vadd(d.data(), &Y[(size_t)j * n], -alpha[j], n);
        }
        const double h0 = ys / yy;
        for (int i = 0; i < n; ++i) d[i] *= h0;
        for (int i = 0; i < bound; ++i) {
            double beta = vdot(&Y[(size_t)j * n], d.data(), n);
            beta /= ysv[j];
            vadd(d.data(), &S[(size_t)j * n], alpha[j] - beta, n);
            j = (j + 1) % m;
        }
        step = 1.0;
        begin_iteration();
    }
};

// ---------------------------------------------------------------------------------------------
// Device-vector solver.  One round = the device executes this candidate's command, evaluates the
// objective at its trial point and returns scalars.
// ---------------------------------------------------------------------------------------------
struct DvCommand {           // host -> device, one per candidate per round (mapped host memory)
    int flags;               // DV_* bits
    int slot;                // history slot to write (s, y) into (the reference's `end` before it is advanced)
    int bound;               // number of history pairs in the two-loop recursion: min(m, k)
    int newest;              // slot index of the newest pair AFTER this advance (= slot)
    double step;             // trial step: x = xp + step * d
};
enum {
    DV_EVAL = 1,             // this candidate takes part in the evaluation of this round
    DV_INIT = 2,             // before the trial: d = -g, xp = x, gp = g                          (lbfgs.hpp:1220, 1262-1263)
    DV_ADVANCE = 4,          // before the trial: history update, two-loop recursion, xp = x, gp = g  (lbfgs.hpp:1354-1411, 1262-1263)
    DV_TRIAL = 8,            // x = xp + step * d                                                   (lbfgs.hpp:825-826)
    DV_RESTORE = 16          // x = xp, g = gp: line search failed for good                         (lbfgs.hpp:1287-1288)
};
struct DvResult {            // device -> host
    double f;                // objective at the trial point
    double dg;               // g . d          (lbfgs.hpp:830)
    double xx, gg;           // x . x, g . g   (lbfgs.hpp:1296-1297)
    double dginit;           // gp . d after INIT / ADVANCE (lbfgs.hpp:756): the first trial of the new search runs in the same round
    double pad[3];
};

class SolverDV {
public:
    void start(int n_, const frx_lbfgs_params &pm_, DvCommand *cmd_) {
        n = n_; pm = pm_; cmd = cmd_;
        m = pm.mem_size;
        ret = lbfgs_check(n, pm);
        evals = 0; k = 0; fx = 0.0;
        pf.assign(pm.past > 0 ? pm.past : 0, 0.0);
        if (ret != 0) { phase = DONE; idle(); return; }
        phase = WAIT_INITIAL;
        cmd->flags = DV_EVAL; cmd->step = 0.0; cmd->slot = cmd->bound = cmd->newest = 0;     // evaluate the start point as is
    }
    bool done() const { return phase == DONE; }
    int status() const { return ret; }
    int iterations() const { return k; }
    int evaluations() const { return evals; }
    double value() const { return fx; }
    // Driver-side guard: a NaN objective passes every comparison of the backtracking search as "accepted" (lbfgs.hpp:986-1010), so
    // with max_iterations = 0 the reference would iterate on NaNs for ever; the drivers stop a candidate after 64 consecutive
    // non-finite values with LBFGSERR_ROUNDING, keeping the last finite point's bookkeeping.
    void give_up(int code) { ret = code; phase = DONE; cmd->flags = 0; cmd->step = 0.0; }
    // the command lives somewhere else from now on (a plan that moves from the per-stage rounds to a cluster of the resident kernel takes its pending command along)
    void rebind(DvCommand *c) { *c = *cmd; cmd = c; }
    bool saw_nonfinite(double f) { nonfinite = (std::isnan(f) || std::isinf(f)) ? nonfinite + 1 : 0; return nonfinite >= 64; }

    void feed(const DvResult &r) {
        ++evals;
        switch (phase) {
        case WAIT_INITIAL: {                                           // lbfgs.hpp:1211-1246
            fx = r.f;
            if (!pf.empty()) pf[0] = fx;
            double xnorm = std::sqrt(r.xx), gnorm = std::sqrt(r.gg);
            if (xnorm < 1.0) xnorm = 1.0;
            if (gnorm / xnorm <= pm.g_epsilon) { finish(LB_ALREADY_MINIMIZED); return; }
            step = 1.0 / std::sqrt(r.gg);                              // 1 / |d|, d = -g
            k = 1; end = 0;
            stepp = step; fp = fx;
            // d = -g  =>  gp . d = -(g . g), known without another round trip
            const int rc = ls.mt_begin(pm, step, fx, -r.gg);
            if (rc != 0) { linesearch_result(rc, true, -r.gg); return; }
            phase = WAIT_MT;
            cmd->flags = DV_EVAL | DV_INIT | DV_TRIAL; cmd->step = ls.step();
            dginit_cur = -r.gg;
            break;
        }
        case WAIT_MT_FIRST: {                                          // first trial of a new search ran together with the advance
            dginit_cur = r.dginit;
            stepp = step; fp = fx;
            const int rc = ls.mt_begin(pm, step, fx, r.dginit);
            if (rc != 0) { linesearch_result(rc, true, r.dginit); return; }
            phase = WAIT_MT;
        }   // fall through: the trial at step 1.0 has already been evaluated
        // FALLTHROUGH
        case WAIT_MT: {
            const double fprev = fx;
            fx = r.f;
            const int rc = ls.mt_feed(pm, r.f, r.dg);
            (void)fprev;
            if (rc == LineSearch::PENDING) { cmd->flags = DV_EVAL | DV_TRIAL; cmd->step = ls.step(); }
            else { xx = r.xx; gg = r.gg; linesearch_result(rc, true, dginit_cur); }
            break;
        }
        case WAIT_BT: {
            fx = r.f;
            const int rc = ls.bt_feed(pm, r.f, r.dg);
            if (rc == LineSearch::PENDING) { cmd->flags = DV_EVAL | DV_TRIAL; cmd->step = ls.step(); }
            else { xx = r.xx; gg = r.gg; linesearch_result(rc, false, dginit_cur); }
            break;
        }
        default: break;
        }
    }

private:
    enum Phase { WAIT_INITIAL, WAIT_MT_FIRST, WAIT_MT, WAIT_BT, DONE };
    int n = 0, m = 0;
    frx_lbfgs_params pm;
    DvCommand *cmd = nullptr;
    Phase phase = DONE;
    int ret = 0, k = 0, end = 0, evals = 0, nonfinite = 0;
    double fx = 0, step = 0, stepp = 0, fp = 0, xx = 0, gg = 0, dginit_cur = 0;
    std::vector<double> pf;
    LineSearch ls;

    void idle() { cmd->flags = 0; cmd->step = 0.0; }
    void finish(int code) { ret = code; phase = DONE; idle(); }

    // lbfgs.hpp:1270-1293
    void linesearch_result(int lsr, bool from_mt, double dginit) {
        step = ls.step();
        if (lsr < 0 && from_mt) {
            step = stepp;
            fx = fp;
            const int rc = ls.bt_begin(step, fx, dginit, pm);
            if (rc != 0) { linesearch_result(rc, false, dginit); return; }
            phase = WAIT_BT;
            cmd->flags = DV_EVAL | DV_TRIAL; cmd->step = ls.step();
            return;
        }
        if (lsr < 0) {                                                 // revert to the previous point; no further evaluation
            fx = fp;                                                   // the objective AT the restored point (see Solver)
            ret = lsr; phase = DONE;
            cmd->flags = DV_RESTORE; cmd->step = 0.0;
            return;
        }
        after_linesearch();
    }
    // lbfgs.hpp:1295-1419 (scalar part; the vector part is the DV_ADVANCE command)
    void after_linesearch() {
        double xnorm = std::sqrt(xx), gnorm = std::sqrt(gg);
        if (xnorm < 1.0) xnorm = 1.0;
        if (gnorm / xnorm <= pm.g_epsilon) { finish(LB_CONVERGENCE); return; }
        if (!pf.empty()) {
            if (pm.past <= k) {
                const double rate = (pf[k % pm.past] - fx) / fx;
                if (std::fabs(rate) < pm.delta) { finish(LB_STOP); return; }
            }
            pf[k % pm.past] = fx;
        }
        if (pm.max_iterations != 0 && pm.max_iterations < k + 1) { finish(LBERR_MAXIMUMITERATION); return; }
        const int bound = (m <= k) ? m : k;
        cmd->slot = end; cmd->bound = bound; cmd->newest = end;
        ++k;
        end = (end + 1) % m;
        step = 1.0;                                                    // lbfgs.hpp:1418
        cmd->flags = DV_EVAL | DV_ADVANCE | DV_TRIAL; cmd->step = 1.0; // first Moré–Thuente trial is always at min(max(1, min_step), max_step)
        if (cmd->step < pm.min_step) cmd->step = pm.min_step;
        if (pm.max_step < cmd->step) cmd->step = pm.max_step;
        phase = WAIT_MT_FIRST;
    }
};

} // namespace frx
