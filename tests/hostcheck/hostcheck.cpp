// TEST INFRASTRUCTURE — compiles the product's per-sample math header (fast-racing_amd/csrc/frx_math.hpp)
// for the HOST so its algebra (reverse-mode adjoints instead of the reference's explicit Jacobians) can be
// checked against the oracle on a CPU-only box.  Not part of the product library; never used as a fallback.
#include <vector>
#include "../../fast-racing_amd/csrc/frx_math.hpp"

extern "C" void hostcheck_penalty(int N, int kappa, const double *T, const double *C, const int *hoff, const double *hrec,
                                  const double *pc9_chi4 /* ell[3], margin, vMax, thrMin, thrMax, bdrMax, g, chi[4] */,
                                  int abscissa_accumulate, double *out20 /* N x 20: cost, gdT, gdC[18] */) {
    frx::PenaltyConst pc;
    pc.ell[0] = pc9_chi4[0]; pc.ell[1] = pc9_chi4[1]; pc.ell[2] = pc9_chi4[2];
    pc.safeMargin = pc9_chi4[3];
    pc.vMaxSqr = pc9_chi4[4] * pc9_chi4[4]; pc.thrMinSqr = pc9_chi4[5] * pc9_chi4[5];
    pc.thrMaxSqr = pc9_chi4[6] * pc9_chi4[6]; pc.bdrMaxSqr = pc9_chi4[7] * pc9_chi4[7];
    pc.gAcc = pc9_chi4[8];
    for (int q = 0; q < 4; q++) pc.chi[q] = pc9_chi4[9 + q];
    for (int i = 0; i < N; i++) {
        const double *c = C + 18 * i;
        double *o = out20 + 20 * i;
        // records in device form: (unit normal, n.(p - org) - margin) with org = point of the first half-space
        const int K = hoff[i + 1] - hoff[i];
        const double *org = hrec + 6 * hoff[i] + 3;
        std::vector<double> r4(4 * (size_t)K);
        for (int k = 0; k < K; k++) {
            const double *r = hrec + 6 * (hoff[i] + k);
            r4[4 * k] = r[0]; r4[4 * k + 1] = r[1]; r4[4 * k + 2] = r[2];
            r4[4 * k + 3] = r[0] * (r[3] - org[0]) + r[1] * (r[4] - org[1]) + r[2] * (r[5] - org[2]) - pc.safeMargin;
        }
        for (int v = 0; v < 20; v++) o[v] = 0.0;
        const double step = T[i] / kappa;
        double s1acc = 0.0;
        for (int j = 0; j <= kappa; j++) {
            const double s1 = abscissa_accumulate ? s1acc : step * j;
            const double omg = (j == 0 || j == kappa) ? 0.5 : 1.0;
            const double alpha = 1.0 / kappa * j;
            double adj[12], P, gTa;
            frx::penalty_sample(c, c, s1, omg * step, pc, org, r4.data(), K, adj, P, gTa);
            const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
            const double b0[6] = {1.0, s1, s2, s3, s4, s5};
            const double b1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
            const double b2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
            const double b3[6] = {0.0, 0.0, 0.0, 6.0, 24.0 * s1, 60.0 * s2};
            o[0] += omg * step * P;
            o[1] += alpha * gTa + omg * P / kappa;
            for (int k = 0; k < 6; k++)
                for (int d = 0; d < 3; d++)
                    o[2 + 3 * k + d] += b0[k] * adj[d] + b1[k] * adj[3 + d] + b2[k] * adj[6 + d] + b3[k] * adj[9 + d];
            s1acc += step;
        }
    }
}

// ---- MINCO map in knot form (fast-racing_amd/csrc/frx_minco.hpp), emulated sequentially on the host ----
#include <vector>
#include "../../fast-racing_amd/csrc/frx_minco.hpp"

namespace {
// solve the knot system for all interior knots by PCR, lanes emulated by a loop; rows[k-1] = knot k
void pcr_solve(std::vector<frx::KnotRow> rows, std::vector<double> &v, std::vector<double> &a) {
    const int n = (int)rows.size();
    frx::KnotRow id; frx::knot_row_identity(id);
    std::vector<frx::KnotRow> nxt(n);
    for (int s = 1; s < n; s *= 2) {
        for (int k = 0; k < n; k++) frx::pcr_step(rows[k], k - s >= 0 ? rows[k - s] : id, k + s < n ? rows[k + s] : id, nxt[k]);
        rows.swap(nxt);
    }
    v.assign(3 * n, 0.0); a.assign(3 * n, 0.0);
    for (int k = 0; k < n; k++) frx::pcr_finish(rows[k], &v[3 * k], &a[3 * k]);
}
void build_rows(int N, const double *T, std::vector<frx::KnotRow> &rows) {
    rows.resize(N - 1);
    for (int k = 1; k <= N - 1; k++) frx::knot_row_matrix(T[k - 1], T[k], rows[k - 1]);
}
} // namespace

// forward: q (3 x (N-1) col-major), T[N], head/tail (col-major p|v|a) -> C (6N x 3 row-major)
extern "C" void hostcheck_minco_forward(int N, const double *T, const double *q, const double *head, const double *tail, double *C,
                                        double *Vout, double *Aout) {
    std::vector<double> P(3 * (N + 1)), V(3 * (N + 1)), A(3 * (N + 1));
    for (int x = 0; x < 3; x++) {
        P[x] = head[x]; V[x] = head[3 + x]; A[x] = head[6 + x];
        P[3 * N + x] = tail[x]; V[3 * N + x] = tail[3 + x]; A[3 * N + x] = tail[6 + x];
    }
    for (int k = 1; k < N; k++) for (int x = 0; x < 3; x++) P[3 * k + x] = q[3 * (k - 1) + x];
    if (N > 1) {
        std::vector<frx::KnotRow> rows;
        build_rows(N, T, rows);
        for (int k = 1; k <= N - 1; k++) {
            frx::KnotRow &R = rows[k - 1];
            for (int x = 0; x < 3; x++)
                frx::knot_row_rhs(T[k - 1], T[k], P[3 * k + x] - P[3 * (k - 1) + x], P[3 * (k + 1) + x] - P[3 * k + x], R.r[x], R.r[3 + x]);
            if (k == 1) {           // known head (v,a) moves to the right-hand side
                for (int x = 0; x < 3; x++) { R.r[x] -= R.L[0] * V[x] + R.L[1] * A[x]; R.r[3 + x] -= R.L[2] * V[x] + R.L[3] * A[x]; }
                for (int i = 0; i < 4; i++) R.L[i] = 0.0;
            }
            if (k == N - 1) {
                for (int x = 0; x < 3; x++) { R.r[x] -= R.U[0] * V[3 * N + x] + R.U[1] * A[3 * N + x]; R.r[3 + x] -= R.U[2] * V[3 * N + x] + R.U[3] * A[3 * N + x]; }
                for (int i = 0; i < 4; i++) R.U[i] = 0.0;
            }
        }
        std::vector<double> v, a;
        pcr_solve(rows, v, a);
        for (int k = 1; k < N; k++) for (int x = 0; x < 3; x++) { V[3 * k + x] = v[3 * (k - 1) + x]; A[3 * k + x] = a[3 * (k - 1) + x]; }
    }
    for (int i = 0; i < N; i++)
        for (int x = 0; x < 3; x++) {
            double c[6];
            frx::hermite_coeffs(T[i], P[3 * i + x], V[3 * i + x], A[3 * i + x], P[3 * i + 3 + x], V[3 * i + 3 + x], A[3 * i + 3 + x], c);
            for (int k = 0; k < 6; k++) C[(6 * i + k) * 3 + x] = c[k];
        }
    if (Vout) for (int i = 0; i < 3 * (N + 1); i++) { Vout[i] = V[i]; Aout[i] = A[i]; }
}

// adjoint: given dC = d f/d c (6N x 3 row-major) at the point (T, q): gdT[N] += implicit part, gdQ (3 x (N-1)) += d f / d q
extern "C" void hostcheck_minco_adjoint(int N, const double *T, const double *q, const double *head, const double *tail,
                                        const double *dC, double *gdT, double *gdQ) {
    std::vector<double> C(18 * N), V(3 * (N + 1)), A(3 * (N + 1)), P(3 * (N + 1));
    hostcheck_minco_forward(N, T, q, head, tail, C.data(), V.data(), A.data());
    for (int x = 0; x < 3; x++) { P[x] = head[x]; P[3 * N + x] = tail[x]; }
    for (int k = 1; k < N; k++) for (int x = 0; x < 3; x++) P[3 * k + x] = q[3 * (k - 1) + x];
    std::vector<double> pb(3 * (N + 1), 0.0), vb(3 * (N + 1), 0.0), ab(3 * (N + 1), 0.0);
    for (int i = 0; i < N; i++)
        for (int x = 0; x < 3; x++) {
            double cb[6], db[6], hb;
            for (int k = 0; k < 6; k++) cb[k] = dC[(6 * i + k) * 3 + x];
            frx::hermite_adjoint(T[i], P[3 * i + x], V[3 * i + x], A[3 * i + x], P[3 * i + 3 + x], V[3 * i + 3 + x], A[3 * i + 3 + x], cb, db, hb);
            pb[3 * i + x] += db[0]; vb[3 * i + x] += db[1]; ab[3 * i + x] += db[2];
            pb[3 * i + 3 + x] += db[3]; vb[3 * i + 3 + x] += db[4]; ab[3 * i + 3 + x] += db[5];
            gdT[i] += hb;
        }
    std::vector<double> muv(3 * (N + 1), 0.0), mua(3 * (N + 1), 0.0);
    if (N > 1) {
        std::vector<frx::KnotRow> rows;
        build_rows(N, T, rows);
        for (int k = 1; k <= N - 1; k++) {
            frx::KnotRow &R = rows[k - 1];
            for (int x = 0; x < 3; x++) { R.r[x] = vb[3 * k + x]; R.r[3 + x] = ab[3 * k + x]; }
            if (k == 1) for (int i = 0; i < 4; i++) R.L[i] = 0.0;
            if (k == N - 1) for (int i = 0; i < 4; i++) R.U[i] = 0.0;
        }
        std::vector<double> v, a;
        pcr_solve(rows, v, a);
        for (int k = 1; k < N; k++) for (int x = 0; x < 3; x++) { muv[3 * k + x] = v[3 * (k - 1) + x]; mua[3 * k + x] = a[3 * (k - 1) + x]; }
    }
    for (int i = 0; i < N; i++)
        for (int x = 0; x < 3; x++) {
            double hb = 0.0;
            const double dlb = frx::knot_adjoint_piece(T[i], P[3 * i + 3 + x] - P[3 * i + x], V[3 * i + x], A[3 * i + x], V[3 * i + 3 + x],
                                                       A[3 * i + 3 + x], muv[3 * i + x], mua[3 * i + x], muv[3 * i + 3 + x], mua[3 * i + 3 + x], hb);
            gdT[i] += hb;
            pb[3 * i + 3 + x] += dlb; pb[3 * i + x] -= dlb;
        }
    for (int k = 1; k < N; k++) for (int x = 0; x < 3; x++) gdQ[3 * (k - 1) + x] += pb[3 * k + x];
}

// ---- SolverDV (device-vector L-BFGS control logic) driven by a HOST emulation of the device commands ----
// The emulation executes DV_INIT / DV_ADVANCE / DV_TRIAL / DV_RESTORE with the same sequential loops as the
// host-vector Solver, so the iterates must be bit-identical to Solver and to the reference lbfgs.hpp: this pins the
// command protocol (what is sent when, which scalar is consumed when) on a CPU-only box.
#include <cstring>
#include "../../fast-racing_amd/csrc/frx_lbfgs.hpp"

extern "C" int hostcheck_lbfgs_dv(int n, double *x, double *fx_out, double (*fn)(void *, const double *, double *, int), void *inst,
                                  const double *p11, int *iters, int *evals) {
    frx_lbfgs_params pm;
    pm.mem_size = (int)p11[0]; pm.g_epsilon = p11[1]; pm.past = (int)p11[2]; pm.delta = p11[3]; pm.max_iterations = (int)p11[4];
    pm.max_linesearch = (int)p11[5]; pm.min_step = p11[6]; pm.max_step = p11[7]; pm.f_dec_coeff = p11[8]; pm.s_curv_coeff = p11[9];
    pm.xtol = p11[10];
    const int m = pm.mem_size;
    std::vector<double> g(n, 0.0), xp(n, 0.0), gp(n, 0.0), d(n, 0.0), S((size_t)m * n, 0.0), Y((size_t)m * n, 0.0), ysv(m, 0.0), al(m, 0.0);
    auto dot = [n](const double *a, const double *b) { double s = 0.; for (int i = 0; i < n; ++i) s += a[i] * b[i]; return s; };
    auto axpy = [n](double *y, const double *v, double c) { for (int i = 0; i < n; ++i) y[i] += c * v[i]; };
    frx::DvCommand cmd;
    frx::DvResult res;
    std::memset(&res, 0, sizeof(res));
    frx::SolverDV sv;
    sv.start(n, pm, &cmd);
    while (cmd.flags != 0) {
        const frx::DvCommand c = cmd;
        if (c.flags & frx::DV_RESTORE) { std::memcpy(x, xp.data(), sizeof(double) * n); std::memcpy(g.data(), gp.data(), sizeof(double) * n); cmd.flags = 0; break; }
        if (c.flags & frx::DV_INIT) {
            for (int i = 0; i < n; ++i) d[i] = -g[i];
            xp.assign(x, x + n); gp = g;
            res.dginit = dot(gp.data(), d.data());
        }
        if (c.flags & frx::DV_ADVANCE) {
            double *se = &S[(size_t)c.slot * n], *ye = &Y[(size_t)c.slot * n];
            for (int i = 0; i < n; ++i) se[i] = x[i] - xp[i];
            for (int i = 0; i < n; ++i) ye[i] = g[i] - gp[i];
            const double ys = dot(ye, se), yy = dot(ye, ye);
            ysv[c.slot] = ys;
            for (int i = 0; i < n; ++i) d[i] = -g[i];
            int j = (c.slot + 1) % m;
            for (int it = 0; it < c.bound; ++it) {
                j = (j + m - 1) % m;
                al[j] = dot(&S[(size_t)j * n], d.data());
                al[j] /= ysv[j];
                axpy(d.data(), &Y[(size_t)j * n], -al[j]);
            }
            const double h0 = ys / yy;
            for (int i = 0; i < n; ++i) d[i] *= h0;
            for (int it = 0; it < c.bound; ++it) {
                double beta = dot(&Y[(size_t)j * n], d.data());
                beta /= ysv[j];
                axpy(d.data(), &S[(size_t)j * n], al[j] - beta);
                j = (j + 1) % m;
            }
            xp.assign(x, x + n); gp = g;
            res.dginit = dot(gp.data(), d.data());
        }
        if (c.flags & frx::DV_TRIAL) { std::memcpy(x, xp.data(), sizeof(double) * n); axpy(x, d.data(), c.step); }
        if (c.flags & frx::DV_EVAL) {
            res.f = fn(inst, x, g.data(), n);
            res.dg = dot(g.data(), d.data()); res.xx = dot(x, x); res.gg = dot(g.data(), g.data());
            sv.feed(res);
        }
    }
    if (fx_out) *fx_out = sv.value();
    if (iters) *iters = sv.iterations();
    if (evals) *evals = sv.evaluations();
    return sv.status();
}

// LineSearch::first_trial_accepted against the state machine it abbreviates: 1 = the two agree on this input
extern "C" int hostcheck_first_trial(double ftol, double gtol, double min_step, double max_step, double xtol, int max_linesearch,
                                     double step0, double f0, double dginit0, double f, double dg, int *fast, int *full) {
    frx_lbfgs_params pm = {};
    pm.f_dec_coeff = ftol; pm.s_curv_coeff = gtol; pm.min_step = min_step; pm.max_step = max_step; pm.xtol = xtol; pm.max_linesearch = max_linesearch;
    const bool a = frx::LineSearch::first_trial_accepted(pm, step0, f0, dginit0, f, dg);
    frx::LineSearch ls;
    const bool b = ls.mt_begin(pm, step0, f0, dginit0) == 0 && ls.mt_feed(pm, f, dg) == 1;
    if (fast) *fast = a; if (full) *full = b;
    return a == b;
}

// ---- host side of setup() and the reference's initial guess (fast-racing_amd/csrc/frx_host_setup.hpp), without a device ----
// The library builds the same HostCand records inside frx_problem_create and calls the same initial_guess_batch from frx_initial_guess;
// here they run alone so that the initial guess (incl. the nested NLS solves of backwardP) is checked against the oracle on a CPU-only box.
#include "../../fast-racing_amd/csrc/frx_host_setup.hpp"

extern "C" int hostcheck_initial_guess(const frx_config *cfg, int B, const int *coarse_n, const double *ini_state, const double *fin_state,
                                       const int *v_off, const double *v_rec, int *x_off /* B + 1 */, double *x0, int cap) {
    const bool softT = cfg->rho > 0;
    std::vector<frx::HostCand> cand(B);
    int vpoly = 0;
    x_off[0] = 0;
    for (int b = 0; b < B; b++) {
        if (frx::host_cand_init(cand[b], *cfg, softT, coarse_n[b], ini_state + 9 * (size_t)b, fin_state + 9 * (size_t)b, v_off + vpoly, v_rec) != FRX_OK) return -1;
        vpoly += 2 * coarse_n[b] - 1;
        x_off[b + 1] = x_off[b] + cand[b].dimT + cand[b].dimP;
    }
    if (x_off[B] > cap) return -2;
    frx::initial_guess_batch(*cfg, softT, cand, x_off, x0);
    return x_off[B];
}

// ---- round 6: host-side rules of the device layouts (pure functions of the product's headers) ----
#include "../../fast-racing_amd/csrc/frx_device.hpp"
#include "../../fast-racing_amd/csrc/frx_solo_layout.hpp"
// k_lbfgs_pre's geometry and the row stride of its history for vectors of at most n elements: out4 = {E, W, 64 W E, stride}
extern "C" void hostcheck_dv_rows(int n, int tight, int *out4) {
    int E = 0, W = 0, PF = 0;
    frx::dv_geometry(n, &E, &W, &PF);
    out4[0] = E; out4[1] = W; out4[2] = 64 * W * E; out4[3] = E ? (int)frx::dv_row_stride(n, E, W, tight != 0) : 0;
}
// LDS layout of the solo launch (doubles): out8 = {ctl, xs, pw, wq, vs, ev, total, doubles the penalty phase needs from vs on}
extern "C" void hostcheck_solo_lds(int maxN, int maxXb, int maxVb, int maxCN, int nsteps, int ppg, int Kmax, int *out8) {
    const frx::SoloLds L = frx::solo_lds(maxN, maxXb, maxVb, maxCN, nsteps, ppg, Kmax);
    out8[0] = L.ctl; out8[1] = L.xs; out8[2] = L.pw; out8[3] = L.wq; out8[4] = L.vs; out8[5] = L.ev; out8[6] = L.total;
    out8[7] = (ppg < maxN ? ppg : maxN) * (Kmax + 1) * 4 + 256 * frx::SOLO_QS + 2;
}
