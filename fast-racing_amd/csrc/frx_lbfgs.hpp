// Host L-BFGS as a RESUMABLE state machine.
//
// The reference's solver is a blocking call with an evaluation callback
// (lbfgs::lbfgs_optimize + lbfgs_evaluate_t, lbfgs.hpp:1103, :223), which cannot batch: B
// candidates would need B separate device round trips per step.  Here every candidate owns
// one Solver object; the driver collects the points of all candidates that are waiting for
// an objective value, evaluates them in ONE batched device pass, and feeds the values back
// (SURVEY.md §7.1-6).  Between two evaluations a Solver runs exactly the arithmetic of
//   lbfgs_optimize ............ lbfgs.hpp:1103-1444
//   line_search_morethuente ... lbfgs.hpp:730-938   (+ update_trial_interval :520-728)
//   line_search_backtracking .. lbfgs.hpp:940-1033  (fallback when Moré–Thuente fails, :1271-1282)
// in the same operation order, so iterates are bit-identical to the blocking solver
// (tests/test_lbfgs.py checks that against the reference header itself).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/frx.h"

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

class Solver {
public:
    // x and g are caller-owned storage of n doubles each (the packed, pinned batch arrays):
    // the solver writes the next trial point into x and expects the gradient at that point in g.
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
        case WAIT_MT: mt_after_eval(f); break;
        case WAIT_BT: bt_after_eval(f); break;
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

    // Moré–Thuente locals (lbfgs.hpp:743-752)
    int count = 0, brackt = 0, stage1 = 0, uinfo = 0;
    double stx, fxl, dgx, sty, fy, dgy, finit, dginit, dgtest, width, prev_width, stmin, stmax;
    double *stp = nullptr;

    static double vdot(const double *a, const double *b, int n) {
        double s = 0.;
        for (int i = 0; i < n; ++i) s += a[i] * b[i];
        return s;
    }
    static void vadd(double *y, const double *v, double c, int n) {
        for (int i = 0; i < n; ++i) y[i] += c * v[i];
    }

    // ---- interpolants, lbfgs.hpp:318-406 ----
    static double cubic_min(double u, double fu, double du, double v, double fv, double dv) {
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
    static double cubic_min_bounded(double u, double fu, double du, double v, double fv, double dv, double xmin, double xmax) {
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
    static double quad_min(double u, double fu, double du, double v, double fv) {
        const double a = v - u;
        return u + du / ((fu - fv) / a + du) / 2 * a;
    }
    static double secant_min(double u, double du, double v, double dv) {
        const double a = u - v;
        return v + dv / (dv - du) * a;
    }

    // update_trial_interval, lbfgs.hpp:520-728
    static int update_trial(double &xs, double &fxs, double &dxs, double &ys, double &fys, double &dys,
                            double &t, double &ft, double &dt, double tmin, double tmax, int &br) {
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
        if (fxs < ft) { ys = t; fys = ft; dys = dt; }
        else {
            if (dsign) { ys = xs; fys = fxs; dys = dxs; }
            xs = t; fxs = ft; dxs = dt;
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

    void finish(int code) { ret = code; phase = DONE; }

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
        int r = mt_begin();
        if (r != 0) linesearch_result(r, true);
    }

    // ---- Moré–Thuente, lbfgs.hpp:730-938 ----
    int mt_begin() {
        count = 0; brackt = 0; stage1 = 1; uinfo = 0;
        stp = &step;
        if (*stp <= 0.) return LBERR_INVALIDPARAMETERS;
        dginit = vdot(gp.data(), d.data(), n);
        if (0 < dginit) return LBERR_INCREASEGRADIENT;
        finit = fx;
        dgtest = pm.f_dec_coeff * dginit;
        width = pm.max_step - pm.min_step;
        prev_width = 2.0 * width;
        stx = sty = 0.;
        fxl = fy = finit;
        dgx = dgy = dginit;
        mt_propose();
        return 0;
    }
    void mt_propose() {                       // loop head up to the evaluation, lbfgs.hpp:790-826
        if (brackt) { stmin = stx <= sty ? stx : sty; stmax = stx >= sty ? stx : sty; }
        else        { stmin = stx; stmax = *stp + 4.0 * (*stp - stx); }
        if (*stp < pm.min_step) *stp = pm.min_step;
        if (pm.max_step < *stp) *stp = pm.max_step;
        if ((brackt && ((*stp <= stmin || stmax <= *stp) || pm.max_linesearch <= count + 1 || uinfo != 0)) ||
            (brackt && (stmax - stmin <= pm.xtol * stmax)))
            *stp = stx;
        std::memcpy(x, xp.data(), sizeof(double) * n);
        vadd(x, d.data(), *stp, n);
        phase = WAIT_MT;
    }
    void mt_after_eval(double f) {            // lbfgs.hpp:829-935
        fx = f;
        double dg = vdot(g, d.data(), n);
        const double ftest1 = finit + *stp * dgtest;
        ++count;
        if ((std::isinf(fx) || std::isnan(fx)) || (brackt && ((*stp <= stmin || stmax <= *stp) || uinfo != 0))) {
            linesearch_result(LBERR_ROUNDING, true); return;
        }
        if (*stp == pm.max_step && fx <= ftest1 && dg <= dgtest) { linesearch_result(LBERR_MAXIMUMSTEP, true); return; }
        if (*stp == pm.min_step && (ftest1 < fx || dgtest <= dg)) { linesearch_result(LBERR_MINIMUMSTEP, true); return; }
        if (brackt && (stmax - stmin) <= pm.xtol * stmax) { linesearch_result(LBERR_WIDTHTOOSMALL, true); return; }
        if (pm.max_linesearch <= count) { linesearch_result(LBERR_MAXIMUMLINESEARCH, true); return; }
        if (fx <= ftest1 && std::fabs(dg) <= pm.s_curv_coeff * (-dginit)) { linesearch_result(count, true); return; }
        if (stage1 && fx <= ftest1 &&
            (pm.f_dec_coeff <= pm.s_curv_coeff ? pm.f_dec_coeff : pm.s_curv_coeff) * dginit <= dg)
            stage1 = 0;
        if (stage1 && ftest1 < fx && fx <= fxl) {
            double fm = fx - *stp * dgtest, fxm = fxl - stx * dgtest, fym = fy - sty * dgtest;
            double dgm = dg - dgtest, dgxm = dgx - dgtest, dgym = dgy - dgtest;
            uinfo = update_trial(stx, fxm, dgxm, sty, fym, dgym, *stp, fm, dgm, stmin, stmax, brackt);
            fxl = fxm + stx * dgtest;
            fy = fym + sty * dgtest;
            dgx = dgxm + dgtest;
            dgy = dgym + dgtest;
        } else {
            uinfo = update_trial(stx, fxl, dgx, sty, fy, dgy, *stp, fx, dg, stmin, stmax, brackt);
        }
        if (brackt) {
            if (0.66 * prev_width <= std::fabs(sty - stx)) *stp = stx + 0.5 * (sty - stx);
            prev_width = width;
            width = std::fabs(sty - stx);
        }
        mt_propose();
    }

    // ---- backtracking, lbfgs.hpp:940-1033 ----
    int bt_begin() {
        count = 0;
        stp = &step;
        if (*stp <= 0.) return LBERR_INVALIDPARAMETERS;
        dginit = vdot(gp.data(), d.data(), n);
        if (0 < dginit) return LBERR_INCREASEGRADIENT;
        finit = fx;
        dgtest = pm.f_dec_coeff * dginit;
        bt_propose();
        return 0;
    }
    void bt_propose() {
        std::memcpy(x, xp.data(), sizeof(double) * n);
        vadd(x, d.data(), *stp, n);
        phase = WAIT_BT;
    }
    void bt_after_eval(double f) {
        const double dec = 0.5, inc = 2.1;
        double wd;
        fx = f;
        ++count;
        if (fx > finit + *stp * dgtest) wd = dec;
        else {
            const double dg = vdot(g, d.data(), n);
            if (dg < pm.s_curv_coeff * dginit) wd = inc;
            else if (dg > -pm.s_curv_coeff * dginit) wd = dec;
            else { linesearch_result(count, false); return; }
        }
        if (*stp < pm.min_step) { linesearch_result(LBERR_MINIMUMSTEP, false); return; }
        if (*stp > pm.max_step) { linesearch_result(LBERR_MAXIMUMSTEP, false); return; }
        if (pm.max_linesearch <= count) { linesearch_result(LBERR_MAXIMUMLINESEARCH, false); return; }
        *stp *= wd;
        bt_propose();
    }

    // lbfgs.hpp:1270-1293
    void linesearch_result(int ls, bool from_mt) {
        if (ls < 0 && from_mt) {
            step = stepp;
            fx = fp;
            int r = bt_begin();
            if (r != 0) linesearch_result(r, false);
            return;
        }
        if (ls < 0) {
            std::memcpy(x, xp.data(), sizeof(double) * n);
            std::memcpy(g, gp.data(), sizeof(double) * n);
            finish(ls);
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

} // namespace frx
