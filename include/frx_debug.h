/*
 * frx_debug.h — diagnostics of libfrx.so: traces, in-kernel profiles, self-tests of single kernels.
 *
 * NOT part of the drop-in boundary (include/frx.h is the reference's SE3GCOPTER / cuda_computer boundary plus its neighbours);
 * nothing here is needed to plan a trajectory.  tests/, bench.py and scripts/ use these entries to look inside a plan.
 * Same conventions as frx.h (int status, frx_last_error()).
 */
#ifndef FRX_DEBUG_H
#define FRX_DEBUG_H

#include "frx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic (tests): with FRX_TRACE set in the environment, frx_optimize records for candidate 0 one row per evaluated command
 * {flags, step, f, g.d, gp.d_new, x.x, g.g}; returns the number of rows and copies up to cap_rows of them (7 doubles each). */
int frx_debug_trace(const frx_problem *p, double *out, int cap_rows);
/* Diagnostic: with FRX_RESIDENT_PROF set, the resident kernel runs its instrumented instantiation and leaves 16 counters of 100 MHz
 * ticks per workgroup ([B][G][16], segments RK_P_* of csrc/frx_round_kernel.hpp), then per leader a 16-bin histogram of its waits for a
 * host command ([B][16]: bin k = shorter than 2^k us), then 32 shader-clock stamps of candidate 0's last forward / adjoint bodies;
 * returns the word count, copies up to cap_words. */
int frx_resident_profile(const frx_problem *p, unsigned long long *out, int cap_words);

/* Direction log of the resident round kernel (csrc/frx_round_kernel.hpp).  The kernel computes the L-BFGS direction in the compact
 * (Byrd-Nocedal-Schnabel) form from a register-resident history; the reference computes it with the two-loop recursion
 * (lbfgs.hpp:1381-1411).  frx_debug_direction_log(p, cap_steps, n_cands) makes every later resident plan on the handle record, for the
 * first n_cands candidates and the first cap_steps accepted steps of each, the pair (s, y) the step added to the history, the gradient g
 * the direction was built from and the direction d the cluster returned (cap_steps = 0 switches the log off).
 * frx_debug_direction_log_read copies candidate `cand`'s records: row r = {s[nxp], y[nxp], g[nxp], d[nxp], slot, pair count}
 * (*row_doubles = 4 nxp + 2 doubles, vectors zero-padded from n to nxp); *rows = records written (<= cap_steps); out may be NULL. */
int frx_debug_direction_log(frx_problem *p, int cap_steps, int n_cands);
int frx_debug_direction_log_read(const frx_problem *p, int cand, double *out, int cap_rows, int *rows, int *row_doubles);
/* A candidate that ends with an L-BFGS error on the resident kernel keeps that verdict, like the reference keeps lbfgs_optimize's - except
 * LBFGSERR_INCREASEGRADIENT (a search that starts uphill: only rounding in the compact-form direction can produce it, the two-loop recursion of
 * the per-stage rounds would not), which is planned again on the per-stage rounds by default.  enable = 1: re-run EVERY failed candidate (round 2's
 * behaviour, diagnostic); environment FRX_RESIDENT_RETRY=0|1|t(argeted) overrides.  frx_debug_resident_counts: candidates of the last plan that
 * ended with an L-BFGS error other than the iteration limit on the resident kernel, and how many were re-run. */
int frx_debug_set_resident_retry(frx_problem *p, int enable);
int frx_debug_resident_counts(const frx_problem *p, int *failed, int *retried);
/* Clusters of the last resident plan: = batch size when the batch fitted the chip, fewer when the candidates went through the work queue. */
int frx_debug_resident_clusters(const frx_problem *p, int *clusters);
/* The leader of a cluster runs the host's line-search state machine in step with it and starts on the command it expects (ADVANCE after an
 * accepted trial, or the search's next trial step) before the host's answer arrives; every prediction is checked against the command the
 * host really sent.  out3 = {rounds started on a predicted ADVANCE, rounds started on a predicted trial step, predictions the host's command
 * did not confirm and that were redone} of the last resident plan, summed over the candidates. */
int frx_debug_resident_predictions(const frx_problem *p, unsigned long long *out3);

/* Diagnostic (bench): average microseconds of each stage kernel of an evaluation at x - {forward, penalty, adjoint} - over `reps`
 * back-to-back launches of one kernel at a time, HIP events on the handle's stream. */
int frx_eval_stage_times(frx_problem *p, const double *x, int reps, double *out3_us);
/* One launch per evaluation (fast-racing_amd/csrc/frx_eval_kernel.hpp): frx_objective_eval[_device] run a batch that the chip holds at once - candidates of <= 64
 * pieces, B x (1 + ceil(ceil(N / pieces per wave) / 4)) workgroups <= the device's CU count - as ONE grid of clusters instead of three stage launches, with
 * the same objective bit for bit and the same gradient to <= 1e-10.  set(0) keeps the three launches, set(1) returns to the default, set(2) (tests) is set(1) with members that leave at once, as if
 * they never got a CU, and a 50 us bound on the leader's waits: the evaluation fails with FRX_ERR_TIMEOUT (objective values NaN) and the handle goes on with three launches; frx_debug_eval_fused = workgroups per candidate of the form in use,
 * 0 = one launch per stage.  The environment variable FRX_EVAL_FUSED=0 turns the form off for every handle created afterwards. */
int frx_debug_set_eval_fused(frx_problem *p, int enable);
int frx_debug_eval_fused(const frx_problem *p);
/* The solo form of an evaluation (one workgroup per candidate runs forward map, penalty integral and adjoint in ONE launch; batches larger than the clusters of the
 * one-launch form reach; the per-stage rounds of frx_optimize take it too).  mode 0 = never, 1 = from the handle's batch-size threshold on (default; FRX_EVAL_SOLO_MIN_B),
 * 2 = at every batch size, also where the cluster form would apply.  Results are bit-identical to the three stage launches (replaces the same objectiveFunc,
 * se3gcopter_cpu.hpp:961-1000).  frx_debug_eval_solo = workgroups of the kernel a CU holds if the next evaluation takes the form, 0 = it does not. */
int frx_debug_set_eval_solo(frx_problem *p, int mode);
/* which penalty kernel a stage launch of this handle takes: 0 = k_penalty, 1 = k_penalty_lat, 2 = k_penalty_lat2 (large batches: four-wave workgroups, two-phase transpose) */
int frx_debug_penalty_kernel(const frx_problem *p);
int frx_debug_eval_solo(const frx_problem *p);
/* Diagnostic (bench): average microseconds of one evaluation at x in the form frx_objective_eval_device takes, over `reps` back-to-back evaluations. */
int frx_eval_launch_time(frx_problem *p, const double *x, int reps, double *out_us);
/* Diagnostic: one evaluation at x in the one-launch form with shader-clock stamps of candidate 0's cluster: out64[0..12] forward map and [16..31] adjoint as
 * frx_profile_phases, [40..43] the leader's entry / end of the forward map / end of the adjoint / end, [44..48] wave 0 of the first member: entry, gate seen,
 * granules staged, samples done, partials out. */
int frx_debug_profile_eval_cluster(frx_problem *p, const double *x, long long *out64);

/* Diagnostic: runs one evaluation at x and returns shader-clock stamps taken at the phase boundaries of candidate 0's
 * k_forward_knot (out32[0..6]) and k_backward_knot (out32[16..24]). */
int frx_profile_phases(frx_problem *p, const double *x, long long *out32);

/* Diagnostic: k_lbfgs_pre (device two-loop recursion) against a host two-loop recursion on random histories, and its
 * duration.  geom4 = {doubles/thread, waves, look-ahead rows, pairs per reduction} or NULL for the library's choice. */
int frx_dv_selftest(int device, int n, int B, int m, int iters, const int *geom4, unsigned seed, double *max_rel_err,
                    double *avg_us);

/* The jump-point neighbour tables in the reference's storage order (JPS3DNeib ns[27][3][26], f1/f2[27][3][12]; JPS2DNeib
 * ns[9][2][8], f1/f2[9][2][2]; graph_search.h:72-127), generated from rules instead of spelled out; for the parity test. */
int frx_jps_tables(int *ns3, int *f13, int *f23, int *ns2, int *f12, int *f22);

/* Host CPU budget of a resident plan (frx_api.cpp, host_cpu_share): *budget = CPUs this process may use (affinity mask, cut by the cgroup quota cpu.max);
 * *share = this plan's part of them when LOCAL_WORLD_SIZE ranks of a node (FRX_LOCAL_RANKS overrides) and `extra_plans` further plans of this process
 * (the other shards of a frx_multi job) spin next to it; *mailbox_threads = the service threads a resident plan of `clusters` clusters would start
 * (the caller included): min(one per sixteen clusters, at most four; share - 1), at least one.  FRX_HOST_CPUS overrides the budget (tests). */
int frx_debug_host_cpu_share(int clusters, int extra_plans, double *budget, int *share, int *mailbox_threads);

/* Take-over (frx_api.cpp, TakeOver): a batch too large for the resident round kernel runs as per-stage rounds until no more candidates are left than the chip has
 * clusters for; those continue on the resident kernel with their history as it stands.  frx_debug_taken_over: how many candidates of the last plan finished
 * that way (0: none).  frx_debug_compact_from_history: the host code that rebuilds the resident kernel's dense state for such a candidate - R^-1 by slot
 * ([128][129]), Y^T Y ([128][128]), D = diag(s.y) ([128]) - from history rows S, Y [m][hs] (n significant doubles per row), `bound` valid pairs, the newest
 * in slot `newest` (frx_compact.hpp); exported so that a CPU test can check it against a dense inverse.  FRX_TAKEOVER=0 switches take-overs off. */
int frx_debug_taken_over(const frx_problem *p, int *candidates);
/* Tests: rounds > 0 makes every later plan on the handle start as per-stage rounds and hand ALL its running candidates to the resident kernel after that many
 * rounds, whatever the batch size (a small batch would otherwise never take this path); 0 returns to the library's own rule.  (Until round 5 an environment
 * variable read inside frx_optimize did this - and silently switched the resident path off in any process that inherited it.) */
int frx_debug_set_takeover_at(frx_problem *p, long rounds);
int frx_debug_compact_from_history(int m, int n, int hs, int bound, int newest, const double *S, const double *Y, double *rinv129, double *yy, double *vd);

/* Diagnostic (bench): the shader clock the device sustains under a latency-bound FP64 load (one lone wave per CU on every CU, a dependent FMA chain for `ms`
 * milliseconds): shader cycles per tick of the constant 100 MHz counter, as MHz - minimum, mean and maximum over the CUs' workgroups.  A round of the resident kernel
 * is a chain of dependent instructions: its time is cycles / this clock, which is what differs between the boxes of a pool running the same code object. */
int frx_debug_shader_clock(int device, double ms, double *mhz_min, double *mhz_mean, double *mhz_max);
/* where the resident plan's mailboxes live: out4 = {NUMA node of the command mailbox's page, of the result mailbox's page, the device's NUMA node, the caller's CPU}; -1 = unknown */
int frx_debug_mailbox_numa(const frx_problem *p, int *out4);

#ifdef __cplusplus
}
#endif
#endif /* FRX_DEBUG_H */
