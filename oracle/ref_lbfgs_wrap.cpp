// TEST INFRASTRUCTURE — builds oracle/_ref/libref_lbfgs.so from the reference's OWN solver
// header, compiled where it lies (-I/root/reference/src/plan_manage/include/se3gcopter; the
// header needs only libc).  No reference source is copied into this repo: this
// translation unit only #includes it and exports a C entry point.
#include "lbfgs.hpp"

#include <vector>

namespace {
struct Rec { std::vector<double> fx, step; std::vector<int> ls; };
thread_local Rec *g_rec = nullptr;
thread_local lbfgs::lbfgs_evaluate_t g_fn = nullptr;
int progress_rec(void *, const double *, const double *, const double fx, const double, const double,
                 const double step, int, int, int ls) {
    if (g_rec) { g_rec->fx.push_back(fx); g_rec->step.push_back(step); g_rec->ls.push_back(ls); }
    return 0;
}
lbfgs::lbfgs_parameter_t unpack(const double *p) {
    lbfgs::lbfgs_parameter_t q;
    lbfgs::lbfgs_load_default_parameters(&q);
    q.mem_size = (int)p[0]; q.g_epsilon = p[1]; q.past = (int)p[2]; q.delta = p[3]; q.max_iterations = (int)p[4];
    q.max_linesearch = (int)p[5]; q.min_step = p[6]; q.max_step = p[7]; q.f_dec_coeff = p[8]; q.s_curv_coeff = p[9];
    q.xtol = p[10];
    return q;
}
} // namespace

extern "C" {
// the hook signature expected by orc_set_lbfgs() in gcopter_oracle.cpp
int ref_lbfgs_optimize(int n, double *x, double *fx, lbfgs::lbfgs_evaluate_t fn, void *inst, const double *params11) {
    lbfgs::lbfgs_parameter_t q = unpack(params11);
    return lbfgs::lbfgs_optimize(n, x, fx, fn, nullptr, nullptr, inst, &q);
}
int ref_lbfgs_run(int n, double *x, double *fx, lbfgs::lbfgs_evaluate_t fn, void *inst, const double *params11, int trace_cap,
                  double *trace_fx, double *trace_step, int *trace_ls, int *trace_len) {
    lbfgs::lbfgs_parameter_t q = unpack(params11);
    Rec rec;
    g_rec = &rec;
    int ret = lbfgs::lbfgs_optimize(n, x, fx, fn, nullptr, &progress_rec, inst, &q);
    g_rec = nullptr;
    int L = (int)rec.fx.size() < trace_cap ? (int)rec.fx.size() : trace_cap;
    for (int i = 0; i < L; i++) { trace_fx[i] = rec.fx[i]; trace_step[i] = rec.step[i]; trace_ls[i] = rec.ls[i]; }
    if (trace_len) *trace_len = (int)rec.fx.size();
    return ret;
}
}
