// TEST INFRASTRUCTURE — part of the CPU oracle, never shipped, never on the product path.
//
// Blocking L-BFGS with More–Thuente line search and a backtracking fallback, restated from
// the reference's solver so the oracle can run without /root/reference being present:
//   reference: src/plan_manage/include/se3gcopter/lbfgs.hpp
//     parameters + defaults ............ lbfgs.hpp:18-140
//     cubic / quadratic interpolants ... lbfgs.hpp:296-401
//     update_trial_interval ............ lbfgs.hpp:520-728
//     line_search_morethuente .......... lbfgs.hpp:730-938
//     line_search_backtracking ......... lbfgs.hpp:940-1033
//     lbfgs_optimize ................... lbfgs.hpp:1103-1444
// The arithmetic (operation order included) follows the cited lines so that iterates are
// bit-identical to the reference solver compiled with the same flags; tests/test_lbfgs.py
// checks that against oracle/_ref (the reference header itself, compiled where it lies).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace orc_lbfgs {

struct Params {               // lbfgs.hpp:18-140 (_default_param)
    int mem_size = 8;
    double g_epsilon = 1e-5;
    int past = 0;
    double delta = 1e-5;
    int max_iterations = 0;
    int max_linesearch = 40;
    double min_step = 1e-20;
    double max_step = 1e20;
    double f_dec_coeff = 1e-4;
    double s_curv_coeff = 0.9;
    double xtol = 1e-16;
};

// status codes keep the reference's numeric values (lbfgs.hpp:149-206)
enum {
    CONVERGENCE = 0, STOP = 1, ALREADY_MINIMIZED = 2,
    ERR_UNKNOWN = -1024, ERR_LOGIC, ERR_CANCELED, ERR_INVALID_N, ERR_INVALID_MEMSIZE,
    ERR_INVALID_GEPSILON, ERR_INVALID_TESTPERIOD, ERR_INVALID_DELTA, ERR_INVALID_MINSTEP,
    ERR_INVALID_MAXSTEP, ERR_INVALID_FDECCOEFF, ERR_INVALID_SCURVCOEFF, ERR_INVALID_XTOL,
    ERR_INVALID_MAXLINESEARCH, ERR_OUTOFINTERVAL, ERR_INCORRECT_TMINMAX, ERR_ROUNDING,
    ERR_MINIMUMSTEP, ERR_MAXIMUMSTEP, ERR_MAXIMUMLINESEARCH, ERR_MAXIMUMITERATION,
    ERR_WIDTHTOOSMALL, ERR_INVALIDPARAMETERS, ERR_INCREASEGRADIENT
};

typedef double (*eval_fn)(void *instance, const double *x, double *g, int n);

static inline double dot(const double *a, const double *b, int n) {
    double s = 0.;                                   // lbfgs.hpp:463-471
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}
static inline void axpy(double *y, const double *x, double c, int n) {
    for (int i = 0; i < n; ++i) y[i] += c * x[i];    // lbfgs.hpp:433-441
}

// minimiser of the cubic through (u,fu,du),(v,fv,dv) — lbfgs.hpp:318-338
static inline double cubic_min(double u, double fu, double du, double v, double fv, double dv) {
    double d = v - u;
    double theta = (fu - fv) * 3 / d + du + dv;
    double p = std::fabs(theta), q = std::fabs(du), r = std::fabs(dv);
    double s = p >= q ? p : q;
    s = s >= r ? s : r;
    double a = theta / s;
    double gamm = s * std::sqrt(a * a - (du / s) * (dv / s));
    if (v < u) gamm = -gamm;
    p = gamm - du + theta;
    q = gamm - du + gamm + dv;
    r = p / q;
    return u + r * d;
}
// safeguarded variant — lbfgs.hpp:352-381
static inline double cubic_min2(double u, double fu, double du, double v, double fv, double dv,
                                double xmin, double xmax) {
    double d = v - u;
    double theta = (fu - fv) * 3 / d + du + dv;
    double p = std::fabs(theta), q = std::fabs(du), r = std::fabs(dv);
    double s = p >= q ? p : q;
    s = s >= r ? s : r;
    double a = theta / s;
    double gamm = a * a - (du / s) * (dv / s);
    gamm = gamm > 0 ? s * std::sqrt(gamm) : 0;
    if (u < v) gamm = -gamm;
    p = gamm - dv + theta;
    q = gamm - dv + gamm + du;
    r = p / q;
    if (r < 0. && gamm != 0.) return v - r * d;
    if (a < 0) return xmax;
    return xmin;
}
static inline double quad_min(double u, double fu, double du, double v, double fv) {
    double a = v - u;                                 // lbfgs.hpp:392-394
    return u + du / ((fu - fv) / a + du) / 2 * a;
}
static inline double quad_min2(double u, double du, double v, double dv) {
    double a = u - v;                                 // lbfgs.hpp:404-406
    return v + dv / (dv - du) * a;
}

// lbfgs.hpp:520-728
static inline int update_trial(double &x, double &fx, double &dx, double &y, double &fy, double &dy,
                               double &t, double &ft, double &dt, double tmin, double tmax, int &brackt) {
    int bound;
    int dsign = dt * (dx / std::fabs(dx)) < 0.;
    double mc, mq, newt;
    if (brackt) {
        if (t <= (x <= y ? x : y) || (x >= y ? x : y) <= t) return ERR_OUTOFINTERVAL;
        if (0. <= dx * (t - x)) return ERR_INCREASEGRADIENT;
        if (tmax < tmin) return ERR_INCORRECT_TMINMAX;
    }
    if (fx < ft) {                      // case 1
        brackt = 1; bound = 1;
        mc = cubic_min(x, fx, dx, t, ft, dt);
        mq = quad_min(x, fx, dx, t, ft);
        newt = (std::fabs(mc - x) < std::fabs(mq - x)) ? mc : mc + 0.5 * (mq - mc);
    } else if (dsign) {                 // case 2
        brackt = 1; bound = 0;
        mc = cubic_min(x, fx, dx, t, ft, dt);
        mq = quad_min2(x, dx, t, dt);
        newt = (std::fabs(mc - t) > std::fabs(mq - t)) ? mc : mq;
    } else if (std::fabs(dt) < std::fabs(dx)) {   // case 3
        bound = 1;
        mc = cubic_min2(x, fx, dx, t, ft, dt, tmin, tmax);
        mq = quad_min2(x, dx, t, dt);
        if (brackt) newt = (std::fabs(t - mc) < std::fabs(t - mq)) ? mc : mq;
        else        newt = (std::fabs(t - mc) > std::fabs(t - mq)) ? mc : mq;
    } else {                            // case 4
        bound = 0;
        if (brackt)      newt = cubic_min(t, ft, dt, y, fy, dy);
        else if (x < t)  newt = tmax;
        else             newt = tmin;
    }
    if (fx < ft) { y = t; fy = ft; dy = dt; }
    else {
        if (dsign) { y = x; fy = fx; dy = dx; }
        x = t; fx = ft; dx = dt;
    }
    if (tmax < newt) newt = tmax;
    if (newt < tmin) newt = tmin;
    if (brackt && bound) {
        mq = x + 0.66 * (y - x);
        if (x < y) { if (mq < newt) newt = mq; }
        else       { if (newt < mq) newt = mq; }
    }
    t = newt;
    return 0;
}

struct Ctx { eval_fn fn; void *inst; int n; long evals; };

// lbfgs.hpp:730-938
static inline int ls_more_thuente(int n, double *x, double *f, double *g, double *stp, const double *s,
                                  const double *xp, const double *gp, double stpmin, double stpmax,
                                  Ctx &cd, const Params &pm) {
    int count = 0, brackt = 0, stage1 = 1, uinfo = 0;
    if (*stp <= 0.) return ERR_INVALIDPARAMETERS;
    double dginit = dot(gp, s, n);
    if (0 < dginit) return ERR_INCREASEGRADIENT;
    double finit = *f, dgtest = pm.f_dec_coeff * dginit;
    double width = stpmax - stpmin, prev_width = 2.0 * width;
    double stx = 0., sty = 0., fx = finit, fy = finit, dgx = dginit, dgy = dginit;
    double stmin, stmax, dg;
    for (;;) {
        if (brackt) { stmin = stx <= sty ? stx : sty; stmax = stx >= sty ? stx : sty; }
        else        { stmin = stx; stmax = *stp + 4.0 * (*stp - stx); }
        if (*stp < stpmin) *stp = stpmin;
        if (stpmax < *stp) *stp = stpmax;
        if ((brackt && ((*stp <= stmin || stmax <= *stp) || pm.max_linesearch <= count + 1 || uinfo != 0)) ||
            (brackt && (stmax - stmin <= pm.xtol * stmax)))
            *stp = stx;
        std::memcpy(x, xp, sizeof(double) * n);
        axpy(x, s, *stp, n);
        *f = cd.fn(cd.inst, x, g, cd.n); ++cd.evals;
        dg = dot(g, s, n);
        double ftest1 = finit + *stp * dgtest;
        ++count;
        if ((std::isinf(*f) || std::isnan(*f)) || (brackt && ((*stp <= stmin || stmax <= *stp) || uinfo != 0)))
            return ERR_ROUNDING;
        if (*stp == stpmax && *f <= ftest1 && dg <= dgtest) return ERR_MAXIMUMSTEP;
        if (*stp == stpmin && (ftest1 < *f || dgtest <= dg)) return ERR_MINIMUMSTEP;
        if (brackt && (stmax - stmin) <= pm.xtol * stmax) return ERR_WIDTHTOOSMALL;
        if (pm.max_linesearch <= count) return ERR_MAXIMUMLINESEARCH;
        if (*f <= ftest1 && std::fabs(dg) <= pm.s_curv_coeff * (-dginit)) return count;
        if (stage1 && *f <= ftest1 &&
            (pm.f_dec_coeff <= pm.s_curv_coeff ? pm.f_dec_coeff : pm.s_curv_coeff) * dginit <= dg)
            stage1 = 0;
        if (stage1 && ftest1 < *f && *f <= fx) {
            double fm = *f - *stp * dgtest, fxm = fx - stx * dgtest, fym = fy - sty * dgtest;
            double dgm = dg - dgtest, dgxm = dgx - dgtest, dgym = dgy - dgtest;
            uinfo = update_trial(stx, fxm, dgxm, sty, fym, dgym, *stp, fm, dgm, stmin, stmax, brackt);
            fx = fxm + stx * dgtest; fy = fym + sty * dgtest;
            dgx = dgxm + dgtest;     dgy = dgym + dgtest;
        } else {
            uinfo = update_trial(stx, fx, dgx, sty, fy, dgy, *stp, *f, dg, stmin, stmax, brackt);
        }
        if (brackt) {
            if (0.66 * prev_width <= std::fabs(sty - stx)) *stp = stx + 0.5 * (sty - stx);
            prev_width = width;
            width = std::fabs(sty - stx);
        }
    }
}

// lbfgs.hpp:940-1033
static inline int ls_backtracking(int n, double *x, double *f, double *g, double *stp, const double *s,
                                  const double *xp, const double *gp, double stpmin, double stpmax,
                                  Ctx &cd, const Params &pm) {
    int count = 0;
    const double dec = 0.5, inc = 2.1;
    if (*stp <= 0.) return ERR_INVALIDPARAMETERS;
    double dginit = dot(gp, s, n);
    if (0 < dginit) return ERR_INCREASEGRADIENT;
    double finit = *f, dgtest = pm.f_dec_coeff * dginit, width;
    for (;;) {
        std::memcpy(x, xp, sizeof(double) * n);
        axpy(x, s, *stp, n);
        *f = cd.fn(cd.inst, x, g, cd.n); ++cd.evals;
        ++count;
        if (*f > finit + *stp * dgtest) width = dec;
        else {
            double dg = dot(g, s, n);
            if (dg < pm.s_curv_coeff * dginit) width = inc;
            else if (dg > -pm.s_curv_coeff * dginit) width = dec;
            else return count;
        }
        if (*stp < stpmin) return ERR_MINIMUMSTEP;
        if (*stp > stpmax) return ERR_MAXIMUMSTEP;
        if (pm.max_linesearch <= count) return ERR_MAXIMUMLINESEARCH;
        *stp *= width;
    }
}

struct Trace {                 // optional per-iteration record for the iterate-equality tests
    std::vector<double> fx, step;
    std::vector<int> ls;
};

// lbfgs.hpp:1103-1444 (no stepbound / progress callbacks: the reference passes nullptr for
// both, se3gcopter_cpu.hpp:1249-1256 and :800-807)
static inline int optimize(int n, double *x, double *ptr_fx, eval_fn fn, void *inst, const Params &pm,
                           long *n_evals = nullptr, int *n_iters = nullptr, Trace *trace = nullptr) {
    const int m = pm.mem_size;
    if (n <= 0) return ERR_INVALID_N;
    if (m <= 0) return ERR_INVALID_MEMSIZE;
    if (pm.g_epsilon < 0.) return ERR_INVALID_GEPSILON;
    if (pm.past < 0) return ERR_INVALID_TESTPERIOD;
    if (pm.delta < 0.) return ERR_INVALID_DELTA;
    if (pm.min_step < 0.) return ERR_INVALID_MINSTEP;
    if (pm.max_step < pm.min_step) return ERR_INVALID_MAXSTEP;
    if (pm.f_dec_coeff < 0.) return ERR_INVALID_FDECCOEFF;
    if (pm.s_curv_coeff <= pm.f_dec_coeff || 1. <= pm.s_curv_coeff) return ERR_INVALID_SCURVCOEFF;
    if (pm.xtol < 0.) return ERR_INVALID_XTOL;
    if (pm.max_linesearch <= 0) return ERR_INVALID_MAXLINESEARCH;

    std::vector<double> xp(n, 0.), g(n, 0.), gp(n, 0.), d(n, 0.);
    std::vector<double> S((size_t)m * n, 0.), Y((size_t)m * n, 0.), alpha(m, 0.), ysv(m, 0.);
    std::vector<double> pf(pm.past > 0 ? pm.past : 0, 0.);
    Ctx cd{fn, inst, n, 0};
    int ret, k = 0;
    double fx = cd.fn(cd.inst, x, g.data(), n); ++cd.evals;
    if (!pf.empty()) pf[0] = fx;
    for (int i = 0; i < n; ++i) d[i] = -g[i];
    double xnorm = std::sqrt(dot(x, x, n)), gnorm = std::sqrt(dot(g.data(), g.data(), n));
    if (xnorm < 1.0) xnorm = 1.0;
    if (gnorm / xnorm <= pm.g_epsilon) {
        ret = ALREADY_MINIMIZED;
    } else {
        double step = 1.0 / std::sqrt(dot(d.data(), d.data(), n));
        k = 1;
        int end = 0;
        for (;;) {
            std::memcpy(xp.data(), x, sizeof(double) * n);
            std::memcpy(gp.data(), g.data(), sizeof(double) * n);
            double stepp = step, fp = fx;
            int ls = ls_more_thuente(n, x, &fx, g.data(), &step, d.data(), xp.data(), gp.data(),
                                     pm.min_step, pm.max_step, cd, pm);
            if (ls < 0) {
                step = stepp; fx = fp;
                ls = ls_backtracking(n, x, &fx, g.data(), &step, d.data(), xp.data(), gp.data(),
                                     pm.min_step, pm.max_step, cd, pm);
            }
            if (ls < 0) {
                std::memcpy(x, xp.data(), sizeof(double) * n);
                std::memcpy(g.data(), gp.data(), sizeof(double) * n);
                ret = ls;
                break;
            }
            if (trace) { trace->fx.push_back(fx); trace->step.push_back(step); trace->ls.push_back(ls); }
            xnorm = std::sqrt(dot(x, x, n));
            gnorm = std::sqrt(dot(g.data(), g.data(), n));
            if (xnorm < 1.0) xnorm = 1.0;
            if (gnorm / xnorm <= pm.g_epsilon) { ret = CONVERGENCE; break; }
            if (!pf.empty()) {
                if (pm.past <= k) {
                    double rate = (pf[k % pm.past] - fx) / fx;
                    if (std::fabs(rate) < pm.delta) { ret = STOP; break; }
                }
                pf[k % pm.past] = fx;
            }
            if (pm.max_iterations != 0 && pm.max_iterations < k + 1) { ret = ERR_MAXIMUMITERATION; break; }

            double *s_e = &S[(size_t)end * n], *y_e = &Y[(size_t)end * n];
            for (int i = 0; i < n; ++i) s_e[i] = x[i] - xp[i];
            for (int i = 0; i < n; ++i) y_e[i] = g[i] - gp[i];
            double ys = dot(y_e, s_e, n), yy = dot(y_e, y_e, n);
            ysv[end] = ys;
            int bound = (m <= k) ? m : k;
            ++k;
            end = (end + 1) % m;
            for (int i = 0; i < n; ++i) d[i] = -g[i];
            int j = end;
            for (int i = 0; i < bound; ++i) {
                j = (j + m - 1) % m;
                alpha[j] = dot(&S[(size_t)j * n], d.data(), n);
                alpha[j] /= ysv[j];
                axpy(d.data(), &Y[(size_t)j * n], -alpha[j], n);
            }
            const double h0 = ys / yy;               // vecscale(d, ys / yy, n), lbfgs.hpp:1397
            for (int i = 0; i < n; ++i) d[i] *= h0;
            for (int i = 0; i < bound; ++i) {
                double beta = dot(&Y[(size_t)j * n], d.data(), n);
                beta /= ysv[j];
                axpy(d.data(), &S[(size_t)j * n], alpha[j] - beta, n);
                j = (j + 1) % m;
            }
            step = 1.0;
        }
    }
    if (ptr_fx) *ptr_fx = fx;
    if (n_evals) *n_evals = cd.evals;
    if (n_iters) *n_iters = k;
    return ret;
}

} // namespace orc_lbfgs
