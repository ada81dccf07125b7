/*
 * frx.h — C ABI of the MI355X-native back-end for Fast-Racing's SE(3) MINCO optimiser.
 *
 * This is the drop-in boundary.  Every entry point names the reference
 * interface it replaces; paths are relative to /root/reference/src/plan_manage/:
 *   CPU.hpp = include/se3gcopter/se3gcopter_cpu.hpp     GPU.hpp = include/se3gcopter/se3gcopter_gpu.hpp
 *   cc.cuh  = include/cuda_computer.cuh                 cc.cu   = src/cuda_computer.cu
 *   lbfgs.hpp = include/se3gcopter/lbfgs.hpp
 *
 * Conventions
 *   - plain pointers and sizes only; no exceptions cross the ABI; every call returns an int
 *     status (0 = FRX_OK, <0 = error, see frx_status) and frx_last_error() gives the text.
 *   - all floating point is IEEE binary64, exactly like the reference.
 *   - a problem handle holds a BATCH of B independent candidate trajectories (the reference
 *     has B = 1); candidates may have different piece counts and different
 *     numbers of free variables.  Per-candidate quantities are packed back to back and
 *     addressed through the offset arrays returned by frx_problem_layout().
 *   - 3x3 boundary states are column-major (p | v | a), as Eigen stores the reference's
 *     iniState/finState (se3_node_cpu.cpp:98-99, CPU.hpp:440-442).
 *   - coefficient blocks are (6N x 3) ROW-major per candidate: row 6i+k = coefficient of t^k
 *     of piece i (CPU.hpp:244,260); i.e. piece-major, 18 contiguous doubles per piece.
 *   - "_device" variants take device pointers and a hipStream_t (passed as void*) and are
 *     asynchronous; the plain variants take host pointers and block.
 *   - the library never falls back to a CPU implementation: without a usable HIP device
 *     frx_problem_create() fails with FRX_ERR_NO_DEVICE.
 *   - diagnostics (traces, in-kernel profiles, self-tests of single kernels) are NOT part of this boundary: include/frx_debug.h.
 */
#ifndef FRX_H
#define FRX_H

#ifdef __cplusplus
extern "C" {
#endif

#define FRX_VERSION 100

typedef enum frx_status {
    FRX_OK = 0,
    FRX_ERR_INVALID_ARG = -1,
    FRX_ERR_NO_DEVICE = -2,        /* no usable HIP device (there is no CPU fallback) */
    FRX_ERR_HIP = -3,              /* HIP runtime error later on */
    FRX_ERR_EMPTY_POLYTOPE = -4,   /* a corridor cell or overlap has no vertices: SE3GCOPTER::setup returns false (CPU.hpp:1118-1121) */
    FRX_ERR_CAPACITY = -5,         /* piece count too large for the LDS-resident band (reference silently caps N <= 100, cc.cuh:24) */
    FRX_ERR_ALLOC = -6,            /* host or device allocation failed (at create or later) */
    FRX_ERR_TIMEOUT = -7           /* a bounded wait on the device expired (FRX_ROUND_TIMEOUT_MS, default 5000); the handle stays usable */
} frx_status;

/* Scalar arguments of SE3GCOPTER::setup (CPU.hpp:1076-1092), named after the ROS parameters
 * that feed them (se3_planner.h:61-98, misc/zhangjiajie_params.yaml). */
typedef struct frx_config {
    double rho;             /* Rho        > 0: soft total time, weight of sum(T) (CPU.hpp:1097-1102) */
    double total_t;         /* TotalT     used only when rho <= 0 */
    double grid_res;        /* gridRes    INFINITY at the only call site (MinCoPlan_CPU.cpp:117) */
    int qd_intervals;       /* QdIntervals = kappa, trapezoid intervals per piece */
    int c2_diffeo;          /* UseC2Diffeo */
    double horiz_half_len;  /* HorizHalfLen  ellipsoid semi-axes (x, y) */
    double vert_half_len;   /* VertHalfLen   ellipsoid semi-axis z */
    double safe_margin;     /* SafeMargin */
    double vel_max;         /* VelMax */
    double thr_acc_min;     /* ThrustAccMin */
    double thr_acc_max;     /* ThrustAccMax */
    double body_rate_max;   /* BodyRateMax */
    double grav_acc;        /* GravAcc */
    double penalty_pvtb[4]; /* PenaltyPVTB  weights: corridor, velocity, thrust (both bounds), body rate */
} frx_config;

/* L-BFGS parameters = lbfgs::lbfgs_parameter_t (lbfgs.hpp:18-140), same defaults. */
typedef struct frx_lbfgs_params {
    int mem_size;
    double g_epsilon;
    int past;
    double delta;
    int max_iterations;
    int max_linesearch;
    double min_step;
    double max_step;
    double f_dec_coeff;
    double s_curv_coeff;
    double xtol;
} frx_lbfgs_params;

typedef struct frx_problem frx_problem;

int frx_version(void);
/* Text of the last error raised on this thread (never NULL). */
const char *frx_last_error(void);
/* Number of usable HIP devices (0 when there is none; never fails). */
int frx_device_count(void);

/* Fills p with lbfgs_load_default_parameters (lbfgs.hpp:1040-1043). */
void frx_lbfgs_default_params(frx_lbfgs_params *p);
/* Fills p with the values SE3GCOPTER::optimize uses (CPU.hpp:1243-1247): mem_size 128, past 3,
 * g_epsilon 1e-16, min_step 1e-32, delta = rel_cost_tol, everything else default. */
void frx_lbfgs_gcopter_params(frx_lbfgs_params *p, double rel_cost_tol);

/*
 * Replaces SE3GCOPTER::setup (CPU.hpp:1076-1186) + cuda_computer::setup (cc.cu:413-466) for a
 * batch of B candidates on HIP device `device`.  Everything that is constant during the
 * optimisation (polytopes, index maps, boundary states, parameters) is uploaded ONCE here;
 * the reference re-packs ~257 KB through mapped memory on every evaluation (cc.cu:492-527).
 *
 *   coarse_n[B]            polytopes (= coarse pieces) per candidate
 *   ini_state, fin_state   B x 9 doubles, column-major 3x3 (p | v | a) per candidate
 *   h_off, h_rec           CSR over ALL candidates' H-polytopes in order: polytope m owns the
 *                          half-space records h_off[m] .. h_off[m+1]-1, 6 doubles each
 *                          (outer normal, point) = one column of the reference's 6 x K matrix
 *                          (MinCoPlan_CPU.cpp:93-105).  Normals are re-normalised (CPU.hpp:1116).
 *   v_off, v_rec           CSR over ALL candidates' V-polytopes; candidate b owns 2*coarse_n[b]-1
 *                          of them in the order [cell 0, overlap 0|1, cell 1, ...] (CPU.hpp:1031-1074);
 *                          3 doubles per vertex.  These are the OUTPUT of geoutils::enumerateVs,
 *                          which stays on the caller's side for now; the
 *                          [v0, v_r - v0] re-basing of CPU.hpp:1049 is done inside.
 */
int frx_problem_create(const frx_config *cfg, int device, int B, const int *coarse_n,
                       const double *ini_state, const double *fin_state,
                       const int *h_off, const double *h_rec,
                       const int *v_off, const double *v_rec,
                       frx_problem **out);
/*
 * Same, but the V-polytopes are enumerated here from the H-polytopes (cells and consecutive overlaps), replacing
 * SE3GCOPTER::extractVs -> geoutils::enumerateVs (CPU.hpp:1031-1074, geoutils.hpp:43-149): the caller passes exactly what
 * MavGlobalPlanner::plan passes to setup (MinCoPlan_CPU.cpp:93-123).  Vertices are ordered lexicographically (the
 * reference's order depends on a process-global RNG; any fixed order parameterises the same polytope).
 * Fails with FRX_ERR_EMPTY_POLYTOPE when a cell or an overlap has no interior (setup() == false, CPU.hpp:1118-1121).
 */
int frx_problem_create_from_h(const frx_config *cfg, int device, int B, const int *coarse_n,
                              const double *ini_state, const double *fin_state,
                              const int *h_off, const double *h_rec, frx_problem **out);
/* Vertices (3 doubles each, lexicographic order) of one H-polytope given as K records (outer normal, point). */
int frx_enumerate_vertices(int K, const double *h_rec, double *v_out, int cap, int *nv);

/* Safe-flight-corridor generation, the step upstream of frx_problem_create_from_h.
 * frx_line_segment_dilate = LineSegment3D::dilate(offset) (decomp_util/line_segment.h:31-35 with find_ellipsoid :136-214,
 * DecompBase::find_polyhedron decomp_base.h:63-83, add_local_bbox line_segment.h:47-85, obstacle filter decomp_base.h:35-40):
 * the cell of segment p1-p2 in the point cloud obs[3 n_obs] as (normal, point) records in the reference's order (planes tangent at
 * points that lie ON the final ellipsoid tie at distance 1 and rounding orders them), optionally
 * the inflated ellipsoid (C row-major 3x3, centre d).  Pass h_rec = NULL to query *n_planes.
 * frx_corridor_generate = the corridor loop of MavGlobalPlanner::plan (MinCoPlan_CPU.cpp:37-105): cells chained greedily along
 * path[3 n_path] (segments shorter than max_seg = 4 m and not `blocked`, restart at 4/5 of the in-cell span), floor and ceiling
 * planes z in [0, map_height] appended; output in the CSR layout frx_problem_create takes.  `blocked` replaces
 * MapUtil::isBlocked (map_util.h:417) and may be NULL (nothing blocks). */
typedef int (*frx_blocked_fn)(const double *a, const double *b, void *user);
int frx_line_segment_dilate(const double *p1, const double *p2, const double *bbox, int n_obs, const double *obs, double offset,
                            int cap, int *n_planes, double *h_rec, double *ell_C, double *ell_d);
int frx_corridor_generate(int n_path, const double *path, int n_obs, const double *obs, const double *bbox, double map_height,
                          double max_seg, frx_blocked_fn blocked, void *user, int cap_polys, int cap_planes, int *n_polys,
                          int *h_off, double *h_rec);

/* The same cell on the DEVICE for a batch of segments against one obstacle cloud (csrc/frx_corridor_kernels.hpp): one workgroup per segment,
 * the candidate points compacted in cloud order into LDS, every step of find_ellipsoid / find_polyhedron (decomp_util line_segment.h:136-214,
 * decomp_base.h:63-83) = an arg-min and a filter over them on 256 lanes.  p1, p2: n_seg x 3; h_rec: n_seg x cap_planes x 6 (outer normal, point),
 * n_planes[s] records of it valid (tangent planes in the reference's order, then the six planes of the local box); ell_C (n_seg x 9),
 * ell_d (n_seg x 3) may be NULL.  FRX_ERR_CAPACITY when a cell's local box holds more than 4096 points or needs more than cap_planes planes. */
int frx_dilate_batch(int device, int n_seg, const double *p1, const double *p2, const double *bbox, int n_obs, const double *obs, double offset,
                     int cap_planes, int *n_planes, double *h_rec, double *ell_C, double *ell_d);

/* Result wire format.  frx_traj_to_msg fills the array fields of quadrotor_msgs/PolynomialTrajectory the way
 * MavGlobalPlanner::traj2msg does (se3_planner.cpp:31-58): per piece 6 duration-normalised coefficients per axis, highest
 * power first (Piece::normalizePosCoeffMat, trajectory.hpp:131-141), time[] = durations, order[] = 5 (num_order = 5,
 * mag_coeff = 1, action = ACTION_ADD are constants of that function).  frx_msg_sample evaluates such a message at time t
 * after its start exactly like traj_server (traj_server.cpp:406-456): position, velocity, acceleration, jerk. */
int frx_traj_to_msg(int n_pieces, const double *T, const double *C, double *coef_x, double *coef_y, double *coef_z,
                    double *time, unsigned *order);
int frx_msg_sample(int n_segment, const double *coef_x, const double *coef_y, const double *coef_z, const double *time,
                   const unsigned *order, double t, double *pos, double *vel, double *acc, double *jerk);

/* Path-search front end on the voxel grid, host code (csrc/frx_search.cpp) -- the step of
 * MavGlobalPlanner::plan that produces the polyline frx_corridor_generate takes (MinCoPlan_CPU.cpp:13-35).
 * frx_voxel_map = what JPS::MapUtil<3> holds (map_util.h:48-75, setMap :365): cell (i,j,k) covers
 * origin + [i,i+1) x [j,j+1) x [k,k+1) * res, stored x fastest; 0 = free, 100 = occupied, -1 = unknown (:322-327). */
typedef struct frx_voxel_map {
    double origin[3];
    int dim[3];
    double res;
    const signed char *cells;
} frx_voxel_map;
/* MapUtil::GlobalMapBuild / setObs (map_util.h:86-136): mark the cells holding the cloud's points occupied (points outside
 * the map are dropped).  Returns the number of points inside (>= 0) or a negative frx_status. */
int frx_map_mark_cloud(const double *origin, const int *dim, double res, int n_pts, const double *pts, signed char *cells);
/* MapUtil::isBlocked (map_util.h:395-425) with the signature of frx_blocked_fn: pass it to frx_corridor_generate with
 * user = the frx_voxel_map.  1 when a cell at or above 100 lies on the 0.8-cell ray walk strictly between a and b. */
int frx_map_is_blocked(const double *a, const double *b, void *map);
/* JPS::GraphSearch::plan (graph_search.h:129-161, graph_search.cpp:79-236) on a dense occupancy array cmap (0 = free, > 0 =
 * occupied, x fastest): A* (use_jps = 0; six face neighbours in 3-D, eight in 2-D -- the only mode plan_manage calls,
 * MinCoPlan_CPU.cpp:15) or jump-point search (26 / 8 neighbours) with heuristic eps * Euclidean distance.  dim[2] = 0 selects
 * the 2-D search.  The path comes back goal first, as GraphSearch::getPath() holds it, cell for cell the one the reference
 * returns (same successor order, comparator and heap steps).  *n_path = 0 when the goal is unreachable or max_expand (> 0)
 * expansions were spent; *cost = g of the goal.  FRX_ERR_CAPACITY when the path holds more than cap cells. */
int frx_grid_search(const signed char *cmap, const int *dim, const int *start, const int *goal, double eps, int use_jps,
                    int max_expand, int cap, int *n_path, int *path_xyz, int *n_expanded, double *cost);
/* JPSPlanner<3>::plan (jps_planner.cpp:333-420): cells of start and goal (must be free: *status = 1 / 2), grid search on the
 * occupied/not-occupied view of the map (unknown cells are traversable, updateMap :309-323), *status = -1 when there is no
 * path; then raw_path (cell centres, start first), path (removeCornerPts forwards and backwards, removeLinePts; :54-117) and
 * sample_path (the centres of the cells the simplified path crosses, samplePath :118-148 = getSamplePath()).  Any output may
 * be NULL; each holds up to cap points (3 doubles). */
int frx_jps_plan(const frx_voxel_map *map, const double *start, const double *goal, double eps, int use_jps, int cap, int *n_raw,
                 double *raw_path, int *n_path, double *path, int *n_sample, double *sample_path, int *status, int *n_expanded);
/* The route of MavGlobalPlanner::plan through the gates (MinCoPlan_CPU.cpp:13-35): legs start -> gate 0 -> ... -> goal,
 * each one frx_jps_plan's sample path, spliced with the shared point kept once.  Legs are independent and run on n_threads
 * host threads (0 = one per core).  leg_status[n_gates + 1] as *status above; when a leg fails *n_out = 0 (the reference
 * splices stale samples there). */
int frx_route_plan(const frx_voxel_map *map, const double *start, const double *goal, int n_gates, const double *gates, double eps,
                   int use_jps, int n_threads, int cap, int *n_out, double *path_out, int *leg_status, int *leg_expanded);

/* Asynchronous evaluation on host buffers (blocking and asynchronous variants): frx_objective_eval_async returns once
 * the work is enqueued on the handle's stream, frx_wait completes it and fills f and g (valid until then; one evaluation in flight per
 * handle).  The *_device entry points are asynchronous on the caller's stream by construction. */
int frx_objective_eval_async(frx_problem *p, const double *x, double *f, double *g);
int frx_wait(frx_problem *p);

/* Post-checks of a result (MinCoPlan_CPU.cpp:131-132): the largest speed and the largest acceleration magnitude of every piece,
 * Piece::getMaxVelRate / getMaxAccRate (trajectory.hpp:177-273; Trajectory::getMaxVelRate/getMaxAccRate = the max over pieces).
 * T[n_pieces], C[n_pieces][6][3] as frx_optimize returns them; either output may be NULL. */
int frx_traj_max_rates(int n_pieces, const double *T, const double *C, double *max_vel, double *max_acc);

/* Replaces ~cuda_computer / kill_kernel (cc.cu:44-49, 566-579; GPU.hpp:907-909). */
void frx_problem_destroy(frx_problem *p);

/* Where the O(n) state of L-BFGS lives during frx_optimize.  In both modes every decision (Moré–Thuente / backtracking
 * line search, convergence and stop tests, error codes: lbfgs.hpp:730-1033, 1295-1352) is taken on the host.
 *   FRX_LBFGS_DEVICE_VECTORS (default) x, g, d, the (s,y) history and the two-loop recursion (lbfgs.hpp:1354-1411) stay on the
 *                            device; per round and candidate 32 B of command go down and 40 B of scalars come back.
 *   FRX_LBFGS_HOST_VECTORS   the vectors live on the host and run in the reference's exact operation order (bit-identical
 *                            iterates given identical f, g); the gradient crosses PCIe every round.
 * The environment variable FRX_LBFGS=host|device overrides the default. */
#define FRX_LBFGS_DEVICE_VECTORS 0
#define FRX_LBFGS_HOST_VECTORS 1
int frx_problem_set_lbfgs_mode(frx_problem *p, int mode);

/* How the MINCO map (q,T)->c and its adjoint are evaluated on the device.  Both compute the same spline:
 *   FRX_SOLVER_KNOT_PCR  (default) quintic-Hermite knot form, SPD 2x2-block tridiagonal system in the knot (v,a),
 *                        parallel cyclic reduction: O(log N) depth (fast-racing_amd/csrc/frx_minco.hpp)
 *   FRX_SOLVER_BANDED_LU the reference's own elimination order: 6N x 6N band, no-pivot LU, solve, solveAdj
 *                        (trajectory.hpp:655-751); sequential in 6N, kept as an on-device cross-check. */
#define FRX_SOLVER_KNOT_PCR 0
#define FRX_SOLVER_BANDED_LU 1
int frx_problem_set_solver(frx_problem *p, int solver);

/* Totals: out6 = {B, total fine pieces, total coarse pieces, total free variables, max half-spaces per piece,
 * sum over fine pieces of their half-space count}. */
int frx_problem_totals(const frx_problem *p, int *out6);
/* Offsets (each B+1 ints): fine pieces, coarse pieces, free variables; plus dimFreeT per candidate (B ints).
 * Candidate b's variables are x[x_off[b] .. x_off[b+1]) = (tau[dim_t[b]], xi[...]) as in CPU.hpp:970-971. */
int frx_problem_layout(const frx_problem *p, int *piece_off, int *coarse_off, int *x_off, int *dim_t);

/* First half of SE3GCOPTER::optimize (CPU.hpp:1237-1240): setInitial + backwardT + backwardP.
 * Host work (tiny NLS solves, CPU.hpp:777-813); x0 has total-free-variables doubles. */
int frx_initial_guess(frx_problem *p, double *x0);

/*
 * Replaces SE3GCOPTER::objectiveFunc (CPU.hpp:961-1000) for the whole batch: x -> (f, grad).
 * One call = steps 1-7 of SURVEY.md §3.4 for every candidate, entirely on the device.
 *   x, g : total-free-variables doubles;  f : B doubles.
 * On the device this is ONE kernel launch (clusters of workgroups, one per candidate) when every candidate has <= 64 pieces and the chip holds the whole batch at
 * once, three stage launches otherwise; the _device form is a pure sequence of launches on the caller's stream (capturable in a hipGraph), one evaluation in flight
 * per handle.  Should a wait inside the one-launch form expire (a device shared with other grids: INTEGRATION.md 8) the objective values concerned are NaN and the
 * blocking form returns FRX_ERR_TIMEOUT; the handle then continues with the three launches.
 */
int frx_objective_eval(frx_problem *p, const double *x, double *f, double *g);
int frx_objective_eval_device(frx_problem *p, const double *x_dev, double *f_dev, double *g_dev, void *hip_stream);
/* The _device form has no host-synchronous point of its own.  After the caller has synchronised its stream: FRX_OK, or FRX_ERR_TIMEOUT when a wait inside a
 * one-launch evaluation expired since the last check (objective values NaN from that evaluation on; the word is cleared and the handle continues with three
 * launches per evaluation).  Direct calls notice by themselves - the first frx_objective_eval_device after the failure became visible takes the three launches -
 * but a captured hipGraph replays what was captured: a caller that replays graphs polls this (or the NaN objective values). */
int frx_eval_status(frx_problem *p);

/*
 * Replaces cuda_computer::compute (cc.cuh:118-134, cc.cu:469-563) = MINCO_S3::addTimeIntPenalty
 * (CPU.hpp:188-408) for the whole batch, with the same ACCUMULATING semantics (cc.cu:551-558):
 *   cost[b] += penalty,  gdT[piece] += d/dT,  gdC[piece*18 ..] += d/dc.
 *   T    : total-fine-pieces doubles (piece durations)
 *   C    : total-fine-pieces x 18 doubles (piece-major coefficients)
 * The _device variant OVERWRITES per-piece partials out_dev[piece*20 + {0: cost, 1: gdT, 2..19: gdC}]
 * and leaves the accumulation to the caller.
 */
int frx_penalty_eval(frx_problem *p, const double *T, const double *C, double *cost, double *gdT, double *gdC);
int frx_penalty_eval_device(frx_problem *p, const double *T_dev, const double *C_dev, double *out_dev, void *hip_stream);
/*
 * The inner boundary ON ITS OWN: a handle built from exactly what cuda_computer::compute receives on every call (cc.cuh:118-134, call site GPU.hpp:219-227) -
 * idxHs, cfgHs, ellipsoid, safeMargin, the limits, the weights ci, cons = cfg->qd_intervals for every piece (GPU.hpp:963-964) - for a caller that keeps the
 * reference's MINCO_S3 / SE3GCOPTER on the host and swaps only `class cuda_computer` (oracle/frx_dropin/cuda_computer.cuh is that class; the reference's
 * se3gcopter_gpu.hpp compiles against it unmodified, tests/test_reference_gpu_header.py).  Polytopes and parameters go up ONCE here instead of through mapped
 * memory on every call (cc.cu:492-527).
 *   piece_n[B]             pieces per candidate
 *   piece_poly[sum pieces] index m of every piece's polytope in the CSR below (= idxHs)
 *   h_off, h_rec           CSR of the H-polytopes, 6 doubles per half-space (outer normal, point) = one column of cfgHs[m]
 * Such a handle serves frx_penalty_eval and frx_penalty_eval_device; every entry point that needs variables or waypoint polytopes returns FRX_ERR_INVALID_ARG.
 */
int frx_penalty_problem_create(const frx_config *cfg, int device, int B, const int *piece_n, const int *piece_poly,
                               const int *h_off, const double *h_rec, frx_problem **out);

/* x -> piece durations T (total fine pieces) and coefficients C (x 18): forwardT/forwardP + MINCO_S3::generate
 * (CPU.hpp:1258-1262, 425-505).  Either output may be NULL. */
int frx_forward(frx_problem *p, const double *x, double *T, double *C);

/*
 * Replaces SE3GCOPTER::optimize (CPU.hpp:1230-1268) for the whole batch: every candidate runs
 * its own L-BFGS (host, lbfgs.hpp:1103-1444 semantics incl. the Moré–Thuente search and the
 * backtracking fallback); all candidates that need an objective value at a given moment share
 * ONE batched device evaluation.
 *   x        in: start point (from frx_initial_guess or the caller); out: minimiser
 *   C, T     optimised coefficients / durations of the final generate() (CPU.hpp:1262-1263); may be NULL
 *   jerk_cost[B]   what the reference returns (CPU.hpp:1267); may be NULL
 *   objective[B]   final penalised objective (discarded by the reference, CPU.hpp:1242); may be NULL
 *   status[B]      lbfgs_optimize return code per candidate (ignored by the reference, CPU.hpp:1249)
 *   iters[B], evals[B]  iteration / evaluation counts; may be NULL
 */
int frx_optimize(frx_problem *p, const frx_lbfgs_params *params, double *x, double *C, double *T,
                 double *jerk_cost, double *objective, int *status, int *iters, int *evals);

/* Wall-clock split of the last frx_optimize call, milliseconds: out4 = {total, device evaluations
 * (launch + copies + sync), host L-BFGS, evaluation rounds}. */
int frx_optimize_stats(const frx_problem *p, double *out4);

/*
 * How frx_optimize runs a plan on the device.  The reference's device side is ONE kernel that stays resident and is driven through
 * a mailbox in mapped host memory (cc.cu:51-64, 396-405, 454-466, 535-548); the default here has the same shape: the RESIDENT ROUND
 * KERNEL (csrc/frx_round_kernel.hpp) - one launch per plan, a cluster of workgroups per candidate, history of the L-BFGS in
 * registers, per-cluster command/result mailboxes - whenever the batch fits the chip (B x G workgroups <= CUs, mem_size <= 128).
 * A batch of up to a few times that many candidates runs on the same kernel through a WORK QUEUE: as many clusters as the chip holds
 * stay resident, and a cluster whose candidate is finished is handed the next one of the batch instead of leaving.  Still larger
 * batches, and any launch on which a device-side wait expires, run one launch per stage and round (k_lbfgs_pre -> k_forward_knot ->
 * k_penalty -> k_backward_knot) - the faster form for many hundreds of candidates.  frx_problem_set_resident(p, 0) pins a handle to
 * the per-stage rounds (environment: FRX_RESIDENT=0), 1 is the default just described, 2 takes the work queue for every batch that
 * exceeds the chip (environment: FRX_RESIDENT_QUEUE=0|1 forbids / forces the queue).  frx_optimize_path reports what the last plan used
 * (0 = per-stage rounds, G > 0 = resident kernel with G workgroups per candidate) and the resident kernel's device-side status word
 * (0 = clean; otherwise the code of the wait that expired).
 */
int frx_problem_set_resident(frx_problem *p, int enable);
int frx_optimize_path(const frx_problem *p, int *resident_used, unsigned *device_status);
/*
 * Multi-GPU plans (the reference is pinned to device 0, cc.cu:414).  The batch is block-partitioned over the devices:
 * one host thread + one frx_problem handle per device runs its shard exactly like frx_optimize - candidates are independent, an
 * evaluation needs no collective - and the plan ends with the winner exchange over RCCL (ncclCommInitAll over the devices of this
 * process): all-gather of (objective, candidate id), 16 bytes per device, then a broadcast of the winner's 6N x 3 coefficients and
 * N durations from the device that owns it.  librccl.so is loaded on first use.  n_devices = 0 takes every visible device (never
 * more shards than candidates); `devices` may be NULL (0, 1, ...).  When several shards share a physical device (tests on a 1-GPU
 * box; RCCL refuses duplicates), or with FRX_MULTI_COMM=host, the same two operations run through an in-process communicator.
 * Arrays of frx_multi_optimize are packed over the WHOLE batch in candidate order (offsets: frx_multi_layout); C, T, jerk_cost,
 * objective, iters, evals and the winner_* outputs may be NULL.  Failed candidates (status < 0) and non-finite objectives never win,
 * ties go to the lowest id; winner_C / winner_T hold winner_n pieces (18 doubles each / one duration each).
 */
typedef struct frx_multi frx_multi;
int frx_multi_create(const frx_config *cfg, int n_devices, const int *devices, int B, const int *coarse_n, const double *ini_state,
                     const double *fin_state, const int *h_off, const double *h_rec, const int *v_off, const double *v_rec, frx_multi **out);
void frx_multi_destroy(frx_multi *m);
int frx_multi_info(const frx_multi *m, int *n_shards, int *uses_rccl, int *shard_lo, int *shard_device);
int frx_multi_layout(const frx_multi *m, int *piece_off, int *x_off);
int frx_multi_initial_guess(frx_multi *m, double *x0);
int frx_multi_optimize(frx_multi *m, const frx_lbfgs_params *params, double *x, double *C, double *T, double *jerk_cost, double *objective,
                       int *status, int *iters, int *evals, int *winner_id, double *winner_objective, double *winner_C, double *winner_T,
                       int *winner_n);
/* 1 when the last frx_multi_optimize exchanged the winner through RCCL, 0 for the in-process communicator. */
int frx_multi_last_exchange(const frx_multi *m);

/*
 * Host-side solver on its own (used by the CPU tests and by integrators that bring their own
 * objective): minimises `count` independent problems with the batched state machine.  `eval`
 * is called with the ids of the problems that need a value; x/g are the packed arrays
 * (problem i at x_off[i]).  Semantics of each problem = lbfgs::lbfgs_optimize (lbfgs.hpp:1103).
 */
typedef void (*frx_batch_eval_fn)(void *instance, int n_active, const int *active_ids,
                                  const double *x, double *f, double *g);
int frx_lbfgs_minimize_batch(int count, const int *x_off, double *x, double *f_out, int *status, int *iters, int *evals,
                             const frx_lbfgs_params *params, frx_batch_eval_fn eval, void *instance, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* FRX_H */
