#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X back-end (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: one rank per GPU; bench.py starts the ranks ITSELF under torch.distributed.run when no
                                                         launcher did, and refuses to run when fewer than N devices are visible or the launcher's
                                                         WORLD_SIZE is not N;  --multi lib: one process, frx_multi_* drives the N devices)

A "step" is ONE batched cost/gradient evaluation x -> (f, grad f) of the headline workload
(BASELINE.json configs[2]: 32 candidate trajectories x 64 pieces x 16 quadrature intervals,
synthetic Zhangjiajie-like 16-gate corridor) = forward map, penalty integral and adjoint (one launch of k_eval_cluster at this
batch size; three stage kernels for batches the chip does not hold at once) on inputs that are already resident in HBM.  value = constraint-samples/s = ranks * B*N*(kappa+1) * K / t,
t = max over ranks of the barrier-bracketed wall time of the K steps.  Weak scaling: every rank
owns its own batch of 32 candidates (different gate perturbations); the evaluation needs no
collective — the only exchange is the final winner selection, outside the timed region.

The JSON line also carries:
  roofline      the dominant kernel of the timed region.  Batches the chip holds at once are evaluated by ONE launch (k_eval_cluster,
                csrc/frx_eval_kernel.hpp): SURVEY.md §8d's bytes of a full objective evaluation fused on device / its average duration; the
                penalty integrator alone (the kernel §8d prices per piece) is roofline.penalty_integrator.  Other batches: three stage
                launches, and the object is the penalty integrator's: algorithmic bytes (sum over pieces of
                312 + 48 K_i) / its average duration, HIP events on the library's launch stream, measured in this run;
                roofline.evaluation = the whole step against §8d's bytes of an evaluation (penalty bytes + 16 n + 24 sum nv per
                candidate); the two knot kernels with their IMPLEMENTATION traffic (stage buffers, saved multipliers - not
                algorithmic bytes).  Counter figures (HBM traffic, VALU busy) come from the committed counter passes of the same
                gpurun call and are marked "from_profile": true
  cpu_baseline  the CPU oracle (faithful restatement of the reference CPU path) timed on the host cores
  plan_*        full SE(3) plan (frx_optimize from the reference's initial guess, stock OptRelTol)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_PEAK_TFLOPS = 78.6  # FP64 vector peak = half the 157.3 TF FP32 vector peak of the same guide


def cpu_baseline(cands, params, kappa, x_state, x_off, budget_s=12.0):
    """Time the CPU oracle on the host cores: objective evaluations of the same candidates at the same
    state, one candidate per task, as many threads as cores.  Test-infrastructure use of oracle/ (the
    measured thing here IS the baseline, never the product)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as ob
    cores = os.cpu_count() or 1
    # CPUs this process may actually use: the GPU boxes run it under a cgroup quota (cpu.max "1600000 100000" = 16 CPUs' worth on a 256-CPU node) - more
    # threads than that are throttled, so the baseline's `cores` is min(threads, quota), not the thread count
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max": quota = float(q) / float(per)
    except Exception:
        pass
    try: affinity = len(os.sched_getaffinity(0))
    except Exception: affinity = cores
    workers = min(cores, len(cands), 32)                # (bounded: the GPU boxes' 256 logical cores are shared, and a 512-candidate share must not turn the baseline into minutes)
    cands = cands[:workers]                            # one candidate per thread
    oracles = [ob.Oracle(c, params, qd_intervals=kappa) for c in cands]
    xs = [x_state[x_off[b]:x_off[b + 1]].copy() for b in range(len(cands))]

    def work(b, reps):
        o, x = oracles[b], xs[b]
        for _ in range(reps):
            o.objective(x)
        return reps

    t0 = time.perf_counter(); work(0, 3); per_eval = (time.perf_counter() - t0) / 3
    # bounded sample: ~budget_s seconds of wall time
    reps = max(1, int(budget_s * workers / (per_eval * len(cands))))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        done = sum(ex.map(lambda b: work(b, reps), range(len(cands))))
    dt = time.perf_counter() - t0
    samples = done * oracles[0].fine_n * (kappa + 1)
    # one full plan of candidate 0 (stock tolerance), then the whole batch with one candidate per thread
    t0 = time.perf_counter()
    # (bounded: the reference runs L-BFGS without an iteration limit, CPU.hpp:1243-1247, and on an infeasible Monte-Carlo scenario - id 170 -
    # it loops on NaN objectives for ever; 20 000 iterations is more than three times the longest feasible plan.  The batch leg is also bounded in SIZE:
    # one plan per thread, so that a 512-candidate share does not turn the baseline into minutes)
    CPU_PLAN_ITERATION_CAP = 20000
    r = oracles[0].optimize(params["opt_rel_tol"], max_iterations=CPU_PLAN_ITERATION_CAP)
    plan_ms = (time.perf_counter() - t0) * 1e3
    batch = oracles[:workers]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        rs = list(ex.map(lambda o: o.optimize(params["opt_rel_tol"], max_iterations=CPU_PLAN_ITERATION_CAP), batch))
    plan_batch_ms = (time.perf_counter() - t0) * 1e3
    # the same plans once more with the OTHER form of the sample abscissa (step * j, cc.cu:152, against the reference's s1 += step, CPU.hpp:400: one rounding
    # apart per sample): what the reference's optimiser does to a last-bit difference - the CPU-vs-CPU spread printed next to device-vs-CPU
    for o in batch: o.set_abscissa_mode(False)                          # (the oracle's default is the reference's accumulated abscissa)
    with ThreadPoolExecutor(max_workers=workers) as ex:
        rs2 = list(ex.map(lambda o: o.optimize(params["opt_rel_tol"], max_iterations=CPU_PLAN_ITERATION_CAP), batch))
    for o in batch: o.set_abscissa_mode(True)
    cpu_baseline.plans = rs; cpu_baseline.plans_other_rounding = rs2          # for main(): coefficient spread against the device's plans of the SAME candidates
    nb = len(batch)
    return {
        "value": samples / dt, "unit": "constraint-samples/s", "cores": int(max(1, min(workers, affinity, quota if quota else workers))), "node_cores": cores, "threads": workers,
        "cpu_quota": quota, "cpu_affinity": affinity, "kind": "port",
        "note": "port = the CPU restatement of the reference path (oracle/, pinned to the reference's own sources compiled here); the reference itself is NOT timed: "
                "it only compiles against oracle/eigen_shim (eager, index-order stand-in for Eigen) and is 41x slower per evaluation in that form (8091 vs 195 us) - not a fair baseline",
        "sample": f"{done} objective evaluations (x->f,grad) of {len(cands)} candidates of the workload at the bench state, "
                  f"{workers} threads on a node with {cores} logical cores (cgroup quota: {quota if quota else 'none'} CPUs), oracle built -O3 -march=x86-64-v3 -ffp-contract=off",
        "us_per_eval_per_candidate_1thread": per_eval * 1e6,
        "plan_ms_one_candidate_1thread": plan_ms, "plan_evals": int(r["evals"]), "plan_iters": int(r["iters"]),
        # (the CPU leg plans at most 32 candidates, one per thread, whatever the GPU batch is: the field says how many, and the rate is per candidate)
        f"plan_ms_batch_of_{nb}": plan_batch_ms, "plan_batch_threads": workers, "plan_batch_candidates": nb, "plans_per_s": nb / (plan_batch_ms * 1e-3),
        "plan_iteration_cap": CPU_PLAN_ITERATION_CAP, f"plan_objective_min_of_the_first_{nb}": float(min(x["objective"] for x in rs)),
    }


def boundary_calls(frx, sc, prob, cands, params, kappa, x_state, device, with_cpu, reps=200):
    """What the two BOUNDARY calls cost with host buffers, PCIe included: a blocking frx_objective_eval (x up, the evaluation, f and the
    gradient down, one synchronisation) and a blocking frx_penalty_eval (T and C up, the integrator, 20 partials per piece down, the host-side accumulation of
    cuda_computer.cu:551-558) - the call the reference's cuda_computer::compute is (cuda_computer.cu:469-563) - at this run's workload and at BASELINE configs[0]
    (ONE candidate x 64 pieces x the stock kappa = 48: the only shape the reference ever calls compute with, se3gcopter_gpu.hpp:219-227), each next to the CPU oracle's
    time for the same call on one thread (the oracle is the checker being timed as the baseline here, never the product).  `value` stays the HBM-resident step."""
    def time_calls(p, x):
        T, Cf = p.forward(x)
        for _ in range(10): p.objective(x); p.penalty(T, Cf)
        t0 = time.perf_counter()
        for _ in range(reps): p.objective(x)
        t_obj = (time.perf_counter() - t0) / reps * 1e6
        t0 = time.perf_counter()
        for _ in range(reps): p.penalty(T, Cf)
        t_pen = (time.perf_counter() - t0) / reps * 1e6
        return t_obj, t_pen, T, Cf

    def cpu_calls(cs, kap, x, x_off, T, Cf, p_off):
        from oracle import binding as ob
        o = ob.Oracle(cs[0], params, qd_intervals=kap)
        xs = x[x_off[0]:x_off[1]]; Ts = T[p_off[0]:p_off[1]]; Cs = Cf[6 * p_off[0]:6 * p_off[1]]
        n = 20
        for _ in range(2): o.objective(xs); o.penalty(Ts, Cs)
        t0 = time.perf_counter()
        for _ in range(n): o.objective(xs)
        t_obj = (time.perf_counter() - t0) / n * 1e6
        t0 = time.perf_counter()
        for _ in range(n): o.penalty(Ts, Cs)
        t_pen = (time.perf_counter() - t0) / n * 1e6
        return t_obj, t_pen

    out = {"what": "host-buffer (PCIe-inclusive) cost of one blocking boundary call, us; python's ctypes call overhead (~2 us) included", "reps": reps}
    t_obj, t_pen, T, Cf = time_calls(prob, x_state)
    row = {"candidates": prob.B, "objective_eval_blocking": t_obj, "penalty_eval_blocking": t_pen,
           "objective_eval_blocking_samples_per_s": prob.samples() / (t_obj * 1e-6), "penalty_eval_blocking_samples_per_s": prob.samples() / (t_pen * 1e-6)}
    if with_cpu:
        c_obj, c_pen = cpu_calls(cands, kappa, x_state, prob.x_off, T, Cf, prob.piece_off)
        row.update({"cpu_oracle_objective_us_per_candidate_1thread": c_obj, "cpu_oracle_penalty_us_per_candidate_1thread": c_pen,
                    "cpu_oracle_objective_us_whole_batch_1thread": c_obj * prob.B, "cpu_oracle_penalty_us_whole_batch_1thread": c_pen * prob.B})
    out["workload"] = row
    B0, N0, g0, k0 = sc.CONFIGS["plumbing"]
    c0 = [sc.make_candidate(0, N0, g0, perturb_id=0)]
    p0 = frx.Problem(c0, params, device=device, qd_intervals=k0)
    x0 = p0.optimize(params["opt_rel_tol"], max_iterations=60)["x"]
    t_obj, t_pen, T, Cf = time_calls(p0, x0)
    row = {"config": "BASELINE configs[0]: 1 candidate x %d pieces x kappa %d" % (N0, k0), "objective_eval_blocking": t_obj, "penalty_eval_blocking": t_pen,
           "penalty_eval_blocking_samples_per_s": p0.samples() / (t_pen * 1e-6),
           "reference_call": "cuda_computer::compute (cuda_computer.cu:469-563): re-packs ~257 KB through mapped memory, flags its persistent kernel, spins on `done`"}
    if with_cpu:
        c_obj, c_pen = cpu_calls(c0, k0, x0, p0.x_off, T, Cf, p0.piece_off)
        row.update({"cpu_oracle_objective_us_1thread": c_obj, "cpu_oracle_penalty_us_1thread": c_pen})
    p0.close()
    out["configs0_plumbing"] = row
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="headline", help="scenario.CONFIGS key (bench workload = headline)")
    ap.add_argument("--candidates-per-gpu", type=int, default=0, help="override the per-GPU batch (configs quoted over 8 GPUs: 256 -> 32, 4096 -> 512 per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plan", action="store_true")
    ap.add_argument("--large-batch", type=int, default=1024, help="candidates of the extra large-batch k_penalty measurement (0 = skip)")
    ap.add_argument("--multi", choices=("ranks", "lib"), default="ranks",
                    help="N > 1 front end (DESIGN.md 5): 'ranks' = one process per GPU over torch.distributed/RCCL (self-launched when no launcher did it); "
                         "'lib' = ONE process driving frx_multi_* (one host thread + handle per device, ncclCommInitAll inside the library)")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, meet at a barrier, print the job's shape and leave (no GPU work: the CPU test of the self-launch path)")
    args = ap.parse_args()

    import torch                                   # (before libfrx touches the HIP runtime: torch brings its own copy, and the first one loaded wins)
    n_dev_torch = torch.cuda.device_count() if torch.cuda.is_available() else 0
    from frx_import import frx
    from fast_racing_amd import dist as frxdist
    # --gpus N decides the job, not the environment: without a launcher this process starts the N ranks itself (the reference is pinned to device 0,
    # cuda_computer.cu:414); a launcher that started another number of ranks, or fewer visible devices than ranks, is an error - never a silent 1-rank run
    n_dev = min(int(frx.lib().frx_device_count()), n_dev_torch)
    if args.multi == "lib":
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--multi lib is ONE process for all devices: do not start it under a multi-rank launcher")
        if args.gpus > 1 and not os.environ.get("FRX_BENCH_DEVICE") and n_dev < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} --multi lib needs {args.gpus} HIP devices, {n_dev} visible")
    else:
        n_launch = frxdist.ranks_to_launch(args.gpus, os.environ, n_dev if not args.launch_check else max(n_dev, args.gpus))
        if n_launch:
            sys.exit(frxdist.self_launch(n_launch, os.path.abspath(__file__), sys.argv[1:]))

    import numpy as np

    lib_mode = args.multi == "lib" and args.gpus > 1
    world = args.gpus if lib_mode else int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        import torch.distributed as dist
        if world > 1 and not lib_mode:
            dist.init_process_group(backend="gloo"); dist.barrier()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "front_end": args.multi, "self_launched": os.environ.get("FRX_BENCH_SELF_LAUNCHED") == "1" or world == 1}), flush=True)
        if world > 1 and not lib_mode:
            dist.barrier(); dist.destroy_process_group()
        return
    from fast_racing_amd import scenario as sc
    if not torch.cuda.is_available() or n_dev < 1:
        raise SystemExit("bench.py needs a HIP device (libfrx has no CPU fallback)")
    # test knobs (not used by the driver): FRX_BENCH_DEVICE pins every rank to one device and FRX_BENCH_BACKEND=gloo replaces RCCL,
    # so that the N > 1 control flow can be exercised on a 1-GPU box
    if os.environ.get("FRX_BENCH_DEVICE"):
        local_rank = int(os.environ["FRX_BENCH_DEVICE"])
        # More than four ranks on ONE device: their evaluation steps run at the same time, and the one-launch evaluation (clusters whose leader waits for members of the
        # same launch, csrc/frx_eval_kernel.hpp) is sized for a chip of its own - five or more such grids side by side can each hold a part of their first clusters and
        # wait for CUs the others hold (bounded: the launch fails after 2 s).  The control-flow test of such a job takes the three stage launches.
        if world > 4: os.environ["FRX_EVAL_FUSED"] = "0"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 and not lib_mode:
        import torch.distributed as dist
        backend = os.environ.get("FRX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    B, N, gates, kappa = sc.CONFIGS[args.config]
    if args.config in ("perturbed256", "montecarlo4096"): B //= 8          # these two are quoted over 8 GPUs (BASELINE.json configs[3], [4])
    if args.candidates_per_gpu > 0: B = args.candidates_per_gpu
    params = sc.ZHANGJIAJIE
    # weak scaling: rank r owns candidates [r*B, (r+1)*B): perturbations of one scenario ("random gate perturbations"), or
    # independent scenarios for the Monte-Carlo sweep
    if args.config == "montecarlo4096":
        cands = [sc.make_candidate(rank * B + b, N, gates) for b in range(B)]
    else:
        cands = [sc.make_candidate(0, N, gates, perturb_id=rank * B + b) for b in range(B)]
    prob = frx.Problem(cands, params, device=local_rank, qd_intervals=kappa)
    stream = torch.cuda.current_stream().cuda_stream
    # --multi lib: this one process owns every shard.  Shard r (its own 32 candidates, like rank r of the other front end) lives on device r
    # (or all of them on FRX_BENCH_DEVICE: control-flow test on a 1-GPU box); the evaluation step launches on every device before it waits for any
    lib_devs = [int(os.environ["FRX_BENCH_DEVICE"]) if os.environ.get("FRX_BENCH_DEVICE") else r for r in range(world)] if lib_mode else []
    lib_cands = [list(cands)] + [[sc.make_candidate(r * B + b, N, gates) if args.config == "montecarlo4096" else sc.make_candidate(0, N, gates, perturb_id=r * B + b)
                                  for b in range(B)] for r in range(1, world)] if lib_mode else []
    lib_shards = []

    # bench state: the iterate after 60 L-BFGS iterations from the reference's initial guess (untimed)
    x0 = prob.initial_guess()
    warm = prob.optimize(params["opt_rel_tol"], x0=x0, max_iterations=60)
    x_state = warm["x"]
    x_dev = torch.from_numpy(x_state).cuda()
    f_dev = torch.zeros(prob.B, dtype=torch.float64, device="cuda")
    g_dev = torch.zeros(prob.NX, dtype=torch.float64, device="cuda")

    for r in range(1, world if lib_mode else 1):
        pr = frx.Problem(lib_cands[r], params, device=lib_devs[r], qd_intervals=kappa)
        xr = pr.optimize(params["opt_rel_tol"], max_iterations=60)["x"]
        with torch.cuda.device(lib_devs[r]):
            st_r = torch.cuda.Stream(device=lib_devs[r])
            lib_shards.append((pr, torch.from_numpy(xr).cuda(), torch.zeros(pr.B, dtype=torch.float64, device="cuda"), torch.zeros(pr.NX, dtype=torch.float64, device="cuda"), st_r))

    def sync_all():
        torch.cuda.synchronize()
        for d in set(lib_devs): torch.cuda.synchronize(d)

    # The launch stream of the timed steps: a stream of its own (a hipGraph cannot be captured on the default stream); `stream` - the default stream - stays
    # what the other legs of this script use.
    bench_stream = torch.cuda.Stream()

    def step(st=None):
        prob.objective_device(x_dev.data_ptr(), f_dev.data_ptr(), g_dev.data_ptr(), bench_stream.cuda_stream if st is None else st)
        for pr, xd, fd, gd, st_r in lib_shards:
            pr.objective_device(xd.data_ptr(), fd.data_ptr(), gd.data_ptr(), st_r.cuda_stream)

    # The K timed steps are bracketed by barrier + synchronize on both sides (host wall clock of the bracket: `ms_per_step_host_wall`) and timed INSIDE the
    # bracket with HIP events recorded on the launch stream(s) - first launch to last kernel end (at K = 20 the bracket is 0.45 ms and
    # the closing synchronize + barrier, ~40 us of host work, was 10 % of it).  `value` and `ms_per_step` are the event figure, MAX over ranks and streams.
    # The K steps go to the device as ONE hipGraph (3 K kernel nodes captured from the very calls a caller makes - frx_objective_eval_device is a pure
    # sequence of kernel launches on the caller's stream, so it can be captured): the gaps between the three stage kernels of a step and between steps
    # (0.3-0.4 us per launch boundary: 20.5 us per step against 19.1-19.7 as a graph, scripts/r05/graph_probe.py) are the launch path's, not the kernels'.
    # `ms_per_step_direct_launches` is the same K steps launched one by one (FRX_BENCH_GRAPH=0 makes that the timed form; --multi lib has no graph form).
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))]
    for _pr, _xd, _fd, _gd, st_r in lib_shards:
        with torch.cuda.device(st_r.device): ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))

    def record(which):
        ev[0][which].record(bench_stream)
        for i, sh in enumerate(lib_shards): ev[1 + i][which].record(sh[4])

    for _ in range(args.warmup):
        step()
    sync_all(); bench_stream.synchronize()
    graph, graph_note = None, None
    GRAPH_WARM_REPLAYS = 2
    if not lib_mode and os.environ.get("FRX_BENCH_GRAPH", "1") != "0":
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=bench_stream):
                for _ in range(args.steps):
                    step(torch.cuda.current_stream().cuda_stream)
            with torch.cuda.stream(bench_stream):                         # (on the stream the timed replay uses: a replay on another stream is a first replay again)
                for _ in range(GRAPH_WARM_REPLAYS):                       # untimed: the first replay uploads the graph, the second still runs 10 % slow (22.0 against
                    graph.replay()                                       # 19.6-19.8 us per step from the third on, scripts/r05/graph_probe.py): warm-up, reported as such
            bench_stream.synchronize()
        except Exception as e:                                           # a box whose runtime cannot capture: direct launches, and the line says so
            graph, graph_note = None, "hipGraph capture failed (" + repr(e)[:120] + "): direct launches"
    sync_all(); bench_stream.synchronize()
    if dist: dist.barrier()
    sync_all()
    t0 = time.perf_counter()
    record(0)
    if graph is not None:
        with torch.cuda.stream(bench_stream): graph.replay()
    else:
        for _ in range(args.steps):
            step()
    record(1)
    sync_all(); bench_stream.synchronize()
    if dist: dist.barrier()
    sync_all()
    dt_wall = time.perf_counter() - t0
    dt = max(a.elapsed_time(b) for a, b in ev) * 1e-3
    # (diagnostic) the same graph replayed back to back: the timed replay above starts on an EMPTY queue, so its interval carries one graph submission (10-20 us, i.e.
    # 0.5-1 us per step at K = 20: the first replay of every burst measures 19.9-20.6 us per step, the following ones 19.5, scripts/r05/graph_ramp_probe.py)
    dt_b2b = None
    if graph is not None:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        with torch.cuda.stream(bench_stream):
            graph.replay(); evs[0].record(bench_stream)
            for i in range(4): graph.replay(); evs[i + 1].record(bench_stream)
        bench_stream.synchronize()
        dt_b2b = min(evs[i].elapsed_time(evs[i + 1]) for i in range(4)) * 1e-3
    # the same K steps as direct launches (untimed by the contract; reported next to the graph figure)
    e_d0, e_d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e_d0.record(bench_stream)
    for _ in range(args.steps):
        step()
    e_d1.record(bench_stream)
    sync_all(); bench_stream.synchronize()
    dt_direct = e_d0.elapsed_time(e_d1) * 1e-3
    if dist:
        t = torch.tensor([dt, dt_wall, dt_direct], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_wall, dt_direct = float(t[0].item()), float(t[1].item()), float(t[2].item())
    samples_per_step = prob.samples()

    # ---- the job's plan: every rank plans its own 32 candidates at the same time, then the only exchange of the whole job (winner selection) ----
    # Everything that only rank 0 does (stage-kernel times, the per-stage and one-candidate plans, the CPU baseline, ...) runs AFTER the process group is
    # gone: ranks 1 .. N-1 used to sit in a collective - a spinning host thread each, under the boxes' quota of 16 CPUs - while
    # rank 0 ran those legs.
    r = None
    if not args.no_plan:
        if dist: dist.barrier()
        # setup-equivalent host work (SE3GCOPTER::setup incl. H->V enumeration, CPU.hpp:1076-1186, and the first half of optimize:
        # setInitial/backwardT/backwardP, CPU.hpp:1237-1240), timed on a second handle built from the H-polytopes alone
        packed = frx.pack_batch(cands)                                     # (the harness' own repacking of the Python candidates is not set-up work of the library)
        # twice: the first from-H handle of a process may also create the library's sleeping setup pool (threads: ~5 ms, once per process - it is warm already
        # when an earlier leg of this run used it); the second is what every further plan of a planner process pays
        t_s = time.perf_counter()
        p2 = frx.Problem(cands, params, device=local_rank, qd_intervals=kappa, enumerate_v=True, packed=packed)
        t_setup_first = (time.perf_counter() - t_s) * 1e3
        p2.close()
        t_s = time.perf_counter()
        p2 = frx.Problem(cands, params, device=local_rank, qd_intervals=kappa, enumerate_v=True, packed=packed)
        t_setup = (time.perf_counter() - t_s) * 1e3
        t_s = time.perf_counter()
        p2.initial_guess()
        t_guess = (time.perf_counter() - t_s) * 1e3
        p2.close()
        one_device = bool(os.environ.get("FRX_BENCH_DEVICE")) and dist is not None
        if one_device:
            # Control-flow test of the N-rank job on a 1-GPU box (FRX_BENCH_DEVICE): a resident grid needs the whole chip, so the ranks' plans cannot run side
            # by side as they do on N devices - they take turns.  What CAN be reproduced of the N-device job is its HOST side: while one rank plans, every
            # other rank keeps ONE thread spinning (its own mailbox thread would, on a device of its own), under the same CPU quota - the regime VERDICT r4
            # weak #6 asks about (eight ranks, 16 CPUs).  Per-rank times are reported; `value` and `plan_ms` of such a run are not measurements.
            import tempfile
            turn_dir = os.path.join(tempfile.gettempdir(), "frx_bench_turns_" + os.environ.get("TORCHELASTIC_RUN_ID", os.environ.get("MASTER_PORT", "0")))
            os.makedirs(turn_dir, exist_ok=True)
            dist.barrier()
            for turn in range(world):
                flag = os.path.join(turn_dir, f"done_{turn}")
                if rank == turn:
                    r = prob.optimize(params["opt_rel_tol"], x0=x0)
                    open(flag, "w").close()
                else:
                    while not os.path.exists(flag): pass                 # busy on purpose (see above)
                dist.barrier()
            if rank == 0:
                import shutil; shutil.rmtree(turn_dir, ignore_errors=True)
        else:
            r = prob.optimize(params["opt_rel_tol"], x0=x0)
        per_rank = [[r["ms_total"], float(r["rounds"])]]
        if dist:                                                         # the job's plan time is the slowest rank's
            mine = torch.tensor(per_rank[0], dtype=torch.float64, device="cuda")
            allv = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine)
            per_rank = [[float(v[0].item()), float(v[1].item())] for v in allv]
            r["ms_total"] = max(v[0] for v in per_rank) if not one_device else sum(v[0] for v in per_rank)
        r["per_rank"] = per_rank; r["one_device"] = one_device
        # winner selection across ranks (the only exchange in the whole job): all-gather (cost, id), broadcast coefficients
        from fast_racing_amd.dist import select_winner
        ids = np.arange(rank * B, rank * B + B)
        gid, obj, owner, wc, wT = select_winner(
            dist, torch.device("cuda", local_rank), r["objective"], ids,
            lambda i: r["C"][6 * prob.piece_off[i]:6 * prob.piece_off[i + 1]], lambda i: r["T"][prob.piece_off[i]:prob.piece_off[i + 1]], N, local_status=r["status"])
    if dist:
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if rank != 0:
        prob.close()
        return

    # the three stage kernels one at a time (HIP events on the library's launch stream, frx_eval_stage_times): the roofline object is
    # reported for the penalty integrator (the kernel SURVEY.md 8d prices); the two knot kernels get sub-objects with their implementation traffic
    stage_us = prob.stage_times(x_state, reps=max(args.steps, 100))
    # ... and the evaluation in the form the timed steps took: ONE launch of clusters (frx::k_eval_cluster, csrc/frx_eval_kernel.hpp) when the batch fits the chip at
    # once - then THAT kernel is the dominant (the only) kernel of the timed region and the roofline object is reported for it, against SURVEY.md 8d's bytes of a
    # "full objective evaluation fused on device"; the penalty integrator alone (the stage kernel) moves to roofline.penalty_integrator.
    fused_G = prob.eval_fused()
    one_us = min(prob.eval_launch_time(x_state, reps=max(args.steps, 100)) for _ in range(3)) if fused_G else None   # (best of three brackets: a bracket of direct launches also holds the launch path's gaps)
    # batches beyond the clusters' reach in the window where it wins (a GPU's share of the Monte-Carlo config): the SOLO launch - one workgroup per candidate runs the three
    # stage bodies back to back (csrc/frx_solo_kernel.hpp; bit-identical results); timed here in both forms
    solo_wgs = prob.eval_solo()
    solo_us = min(prob.eval_launch_time(x_state, reps=max(args.steps, 100)) for _ in range(3)) if solo_wgs else None
    one1_us = one_us if fused_G else solo_us                           # the ONE kernel of a timed step, where the step is one launch
    prob.set_eval_fused(False)
    prob.set_eval_solo(0)
    three_us = min(prob.eval_launch_time(x_state, reps=max(args.steps, 100)) for _ in range(3))
    prob.set_eval_solo(1)
    prob.set_eval_fused(True)
    # evaluation time along the optimisation (SURVEY.md 8d "kernel-only benchmark state"): the reference initial guess and the iterates
    # after 10 / 20 / 40 / 80 iterations; the headline `value` is taken at the 60-iteration state above
    states = []
    for it in (0, 10, 20, 40, 80):
        xs = x0 if it == 0 else prob.optimize(params["opt_rel_tol"], x0=x0, max_iterations=it)["x"]
        st_us = prob.stage_times(xs, reps=50)
        states.append({"iterations": it, "forward_us": st_us["forward"], "penalty_us": st_us["penalty"], "adjoint_us": st_us["adjoint"],
                       "samples_per_s": samples_per_step / (sum(st_us.values()) * 1e-6)})
    T_dev = torch.zeros(prob.P, dtype=torch.float64, device="cuda")
    C_dev = torch.zeros(prob.P * 18, dtype=torch.float64, device="cuda")
    T_h, C_h = prob.forward(x_state)
    T_dev.copy_(torch.from_numpy(T_h)); C_dev.copy_(torch.from_numpy(C_h.reshape(-1)))
    out_dev = torch.zeros(prob.P * 20, dtype=torch.float64, device="cuda")
    pen_us = stage_us["penalty"]
    alg_bytes = prob.algorithmic_bytes()
    pairs_per_step = (kappa + 1) * prob.sum_K                        # sum over pieces of (kappa+1) K_i (secondary metric)
    achieved = alg_bytes / (pen_us * 1e-6) / 1e9
    # algorithmic bytes of the two knot kernels: per candidate x (8 n), waypoint polytopes (24 per vertex), the stage
    # buffers T, C (152 N), the saved reduction multipliers ((8 steps + 4) 8 N), out20 (160 N), d and g (8 n each)
    nx = np.diff(prob.x_off).astype(np.int64); npc = np.diff(prob.piece_off).astype(np.int64)
    nvert = np.array([sum(v.shape[1] for v in c.v_polys[1::2]) for c in cands], dtype=np.int64)
    steps = np.array([max(int(np.ceil(np.log2(max(n - 1, 1)))), 0) for n in npc], dtype=np.int64)
    fwd_bytes = int(np.sum(8 * nx + 24 * nvert + 152 * npc + (8 * steps + 4) * 8 * npc))
    adj_bytes = int(np.sum(8 * nx + 24 * nvert + 152 * npc + 160 * npc + (8 * steps + 4) * 8 * npc + 8 * nx + 8))
    stage_bytes = {"forward": fwd_bytes, "penalty": alg_bytes, "adjoint": adj_bytes}
    pen_kernel = prob.penalty_kernel()                                 # launch_penalty's choice for this handle (csrc/frx_device.hip): k_penalty_lat, or k_penalty_lat2 from four-wave workgroups on
    # SURVEY.md 8d, "full objective evaluation": penalty bytes + x in and g out (16 n) + the waypoint polytopes (24 bytes per vertex) per candidate
    eval_bytes = int(alg_bytes + np.sum(16 * nx + 24 * nvert))
    # FP64 work of the penalty integrator (scripts/count_fp64.py: FP64 flops per sample counted in the emitted ISA, no corridor violation)
    fp64 = None
    fp64_dyn = None                                                     # dynamic counts of the same gpurun call's counter pass (scripts/r05/gpu_pmc.sh), per launch class
    fc = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r06_fp64_count_k_penalty.json", "r05_fp64_count_k_penalty.json", "r04_fp64_count_k_penalty.json", "r02_fp64_count_k_penalty.json")) if os.path.exists(f)), "")
    if os.path.exists(fc):
        fj = json.load(open(fc))
        fl = fj["flops_per_sample_no_violation"]
        fp64 = {"flops_per_sample": fl, "flops_per_sample_all_corridor_planes_violated": fj["flops_per_sample_all_violated"], "source": os.path.relpath(fc, ROOT),
                "achieved_tflops": fl * samples_per_step / (pen_us * 1e-6) / 1e12, "peak_tflops": FP64_PEAK_TFLOPS,
                "frac": fl * samples_per_step / (pen_us * 1e-6) / 1e12 / FP64_PEAK_TFLOPS}
    # Counter passes (rocprofv3 --pmc, one counter set per pass, scripts/r03/gpu_pmc.sh) of the gpurun call that produced the committed bench line:
    # read from profiles/, never measured inside this process - hence "from_profile".  FETCH_SIZE / WRITE_SIZE are converted to bytes with the
    # factors calibrated in the same call on a coalesced copy of known size (8-byte and 16-byte accesses per lane).
    traffic, traffic_src, valu, knot_traffic, calib, one_traffic = None, None, None, {}, None, None
    pmc_file = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_headline.json", "r05_pmc_headline.json", "r04_pmc_headline.json", "r03_pmc_headline.json")) if os.path.exists(f)), None)
    if pmc_file and args.config == "headline":
        pj = json.load(open(pmc_file))
        traffic, traffic_src, calib = pj["traffic_bytes_per_launch"], os.path.relpath(pmc_file, ROOT), {k: v["bytes_per_counted_byte"] for k, v in pj["calibration"].items()}
        for kname, key in (("k_forward_knot", "forward"), ("k_backward_knot", "adjoint")):
            grids = pj["kernels"].get(kname, {})
            if grids:
                small = sorted(grids, key=lambda k: int(k.split("_")[1]))[0]
                knot_traffic[key] = grids[small].get("traffic_bytes_per_launch_range")
        one_traffic = pj.get("k_eval_cluster", {}).get("traffic_bytes_per_launch_range")
        fd = pj.get("fp64_k_penalty_lat", {})
        if fd and "error" not in fd:
            fcls = sorted(fd.items(), key=lambda kv: int(kv[0].split("_")[1]))
            fp64_dyn = {"headline": fcls[0][1], "large_batch": fcls[-1][1], "source": os.path.relpath(pmc_file, ROOT),
                        "what": "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 / SQ_WAVES of k_penalty_lat at the bench state (a lane is a sample): flops = ADD + MUL + TRANS + 2 FMA"}
            if fp64:
                fl = fcls[0][1]["flops_per_sample"]
                fp64.update({"flops_per_sample": fl, "flops_per_sample_static_count": fp64["flops_per_sample"], "from_profile": True, "dynamic": fp64_dyn,
                             "achieved_tflops": fl * samples_per_step / (pen_us * 1e-6) / 1e12, "frac": fl * samples_per_step / (pen_us * 1e-6) / 1e12 / FP64_PEAK_TFLOPS})
        cls = sorted(pj.get("valu_k_penalty_lat", {}).items(), key=lambda kv: int(kv[0].split("_")[1]))          # launch classes by grid size: headline, large batch
        if cls:
            valu = {"from_profile": True, "source": traffic_src, "headline_valu_busy": cls[0][1]["valu_busy_frac"], "large_batch_valu_busy": cls[-1][1]["valu_busy_frac"],
                    "large_batch_waves_per_simd": cls[-1][1]["mean_waves_per_simd"], "valu_insts_per_wave": cls[-1][1]["valu_insts_per_wave"],
                    "large_batch_non_fp64_valu_frac": cls[-1][1].get("non_fp64_valu_frac"), "large_batch_wait_frac_of_wave_cycles": cls[-1][1].get("wait_frac_of_wave_cycles"),
                    "dynamic_instructions_headline": pj.get("dynamic_instructions_headline")}

    if pmc_file and args.config == "montecarlo4096":
        # the knot kernels at Monte-Carlo scale: counter traffic of the 512-candidate launches of the same record call (kernel_sweep.py --batches 32,512,1024)
        pj = json.load(open(pmc_file))
        traffic_src = os.path.relpath(pmc_file, ROOT)
        for kname, key in (("k_forward_knot", "forward"), ("k_backward_knot", "adjoint")):
            e = pj["kernels"].get(kname, {}).get("grid_%d" % (256 * B))
            if e: knot_traffic[key] = e.get("traffic_bytes_per_launch_range")
        one_traffic = (pj.get("k_eval_solo") or {}).get("grid_%d" % (256 * B), {}).get("traffic_bytes_per_launch_range")   # the timed form at this size: the solo launch
    # the same kernel on a large batch (the headline batch replicated: every replica owns its data in HBM), where the HBM
    # fraction is meaningful; reported next to the headline-size figure, which is launch-latency bound
    large = None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if args.large_batch > 0 and rank == 0:
        rep = max(1, args.large_batch // B)
        big = frx.Problem(cands * rep, params, device=local_rank, qd_intervals=kappa)
        Tb = torch.from_numpy(np.tile(T_h, rep)).cuda(); Cb = torch.from_numpy(np.tile(C_h.reshape(-1), rep)).cuda()
        ob_ = torch.zeros(big.P * 20, dtype=torch.float64, device="cuda")
        for _ in range(5):
            big.penalty_device(Tb.data_ptr(), Cb.data_ptr(), ob_.data_ptr(), stream)
        e0.record()
        for _ in range(30):
            big.penalty_device(Tb.data_ptr(), Cb.data_ptr(), ob_.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        large = {"candidates": big.B, "avg_kernel_us": us, "achieved": big.algorithmic_bytes() / (us * 1e-6) / 1e9, "unit": "GB/s",
                 "frac": big.algorithmic_bytes() / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "kernel_samples_per_s": big.samples() / (us * 1e-6)}
        if fp64:
            fl_big = fp64_dyn["large_batch"]["flops_per_sample"] if fp64_dyn else fp64["flops_per_sample"]
            large["fp64_flops_per_sample"] = fl_big
            large["fp64_frac"] = fl_big * big.samples() / (us * 1e-6) / 1e12 / FP64_PEAK_TFLOPS
        if valu and valu.get("valu_insts_per_wave") and valu.get("large_batch_waves_per_simd"):
            # What THIS instruction count can reach: every SIMD issues one VALU instruction of a wave64 per 4 cycles at best, so a launch of
            # W waves per SIMD with V VALU instructions each lasts at least W V 4 cycles; the HBM fraction at 100 % VALU issue is the kernel's own ceiling
            n_waves = -(-big.P // max(1, (64 * 4) // (kappa + 1))) * 4              # four-wave workgroups of floor(256 / (kappa + 1)) pieces (LaunchGeom::ppg)
            per_simd = n_waves / (256.0 * 4.0)
            t_floor_us = per_simd * valu["valu_insts_per_wave"] * 4.0 / 2400.0      # 2.4 GHz
            large["valu_ceiling_us"] = t_floor_us
            large["valu_ceiling_frac"] = big.algorithmic_bytes() / (t_floor_us * 1e-6) / 1e9 / HBM_PEAK_GBS
            large["valu_ceiling_note"] = ("bytes / (waves per SIMD x VALU instructions per wave x 4 cycles at 2.4 GHz) / HBM peak: the fraction this kernel would reach at 100 %% VALU issue - "
                                          "its own ceiling; %d waves, %.1f per SIMD, %.0f VALU instructions per wave (counter pass %s)" % (n_waves, per_simd, valu["valu_insts_per_wave"], valu.get("source")))
        big.close(); del Tb, Cb, ob_

    # the HBM-bound kernel of the path: the L-BFGS two-loop recursion (k_lbfgs_pre) streams every candidate's (s, y) history twice
    # per accepted step.  Driven alone by frx_dv_selftest (HIP events around each launch) at the headline vector length.
    hbm_kernel = None
    if rank == 0 and args.large_batch > 0:
        n_x = int(np.max(np.diff(prob.x_off))); m_hist = 128; bl = 256
        row = min(64 * w * e for e in (2, 4, 6, 8) for w in range(1, 9) if 64 * w * e >= n_x)   # the geometry's row, frx::dv_geometry (768 at n = 641) ...
        tight = (n_x + 2 + 15) // 16 * 16                                  # ... stored as n + 2 rounded up to 16 doubles since round 6 (frx::dv_row_stride: 656 at n = 641)
        if row >= 512 and tight < row: row = tight
        err, us = frx.dv_selftest(n_x, B=bl, m=m_hist, iters=160)
        byts = bl * m_hist * 2 * 2 * row * 8                               # S and Y rows, read once in each of the two loops
        hbm_kernel = {"kernel": "frx::k_lbfgs_pre", "candidates": bl, "history_pairs": m_hist, "vector_length": n_x, "avg_kernel_us": us,
                      "history_row_doubles": row, "bytes_per_launch": byts, "achieved": byts / (us * 1e-6) / 1e9, "unit": "GB/s", "frac": byts / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                      "check_rel_err_vs_host_recursion": err}
        if pmc_file and "k_lbfgs_pre" in json.load(open(pmc_file)):         # counter pass of the same record call (scripts/r04/gpu_pmc.sh), calibrated in that call
            hbm_kernel["traffic"] = json.load(open(pmc_file))["k_lbfgs_pre"]["traffic_bytes_per_launch"]; hbm_kernel["traffic_source"] = os.path.relpath(pmc_file, ROOT); hbm_kernel["traffic_from_profile"] = True

    plan = {}
    if not args.no_plan:
        if rank == 0:
            # the same plan with one launch per stage and round (the round-1 path), and the reference's real use: ONE candidate
            prob.set_resident(False)
            r_ps = prob.optimize(params["opt_rel_tol"], x0=x0)
            prob.set_resident(True)
            r_q = None
            if r["resident"] == 0 or r["clusters"] < B or r.get("taken_over"):
                # a batch larger than the chip holds at once (Monte-Carlo share: 512 per GPU): the resident kernel's work queue, whatever
                # the default chose for this size - both paths on the line
                prob.set_resident(2)
                r_q = prob.optimize(params["opt_rel_tol"], x0=x0)
                prob.set_resident(True)
            # the reference's real use is ONE candidate: SE3GCOPTER::setup + optimize of MinCoPlan_CPU.cpp:113-126, its timer around both
            packed1 = frx.pack_batch(cands[:1])
            t_s = time.perf_counter()
            p1 = frx.Problem(cands[:1], params, device=local_rank, qd_intervals=kappa, enumerate_v=True, packed=packed1)
            t_setup1 = (time.perf_counter() - t_s) * 1e3
            t_s = time.perf_counter()
            x01 = p1.initial_guess()
            t_guess1 = (time.perf_counter() - t_s) * 1e3
            r_b1 = p1.optimize(params["opt_rel_tol"], x0=x01)
            p1.close()
        r_lib = None
        if lib_mode:
            # the whole job through the library's own multi-device front end: one host thread + handle per device, winner over RCCL (ncclCommInitAll)
            mp_ = frx.MultiProblem([c for sh in lib_cands for c in sh], params, devices=lib_devs, qd_intervals=kappa)
            xm = mp_.initial_guess()
            mp_.optimize(params["opt_rel_tol"], x0=xm, max_iterations=30)           # warm: communicator, resident kernels
            t_s = time.perf_counter()
            r_lib = mp_.optimize(params["opt_rel_tol"], x0=xm)
            r_lib["ms_wall"] = (time.perf_counter() - t_s) * 1e3
            r_lib["n_shards"], r_lib["uses_rccl"] = mp_.n_shards, mp_.uses_rccl
            mp_.close()
        plan = {"plan_setup_ms": t_setup, "plan_setup_ms_first_handle_of_this_kind_in_the_process": t_setup_first, "plan_initial_guess_ms": t_guess, "plan_ms_with_setup": r["ms_total"] + t_setup + t_guess,
                "plan_lbfgs_mode": os.environ.get("FRX_LBFGS", "device"), "plan_ms": r["ms_total"], "plan_ms_device": r["ms_device"], "plan_ms_host_lbfgs": r["ms_host"],
                "plan_rounds": r["rounds"], "plan_iters_max": int(r["iters"].max()), "plan_evals_max": int(r["evals"].max()),
                "plan_status_ok": int(np.sum(r["status"] >= 0)), "plan_objective_min": float(r["objective"].min()),
                "plan_resident_failed": int(r["resident_failed"]), "plan_resident_retried": int(r["resident_retried"]),
                "plan_path": ("one launch per stage and round until %d candidates were left, which finished on the resident round kernel (take-over, %d workgroups per candidate)" % (r["taken_over"], r["resident"])) if r.get("taken_over") else
                             ("resident round kernel, %d workgroups per candidate, %d clusters%s" % (r["resident"], r["clusters"], " (work queue)" if r["clusters"] < B else "")) if r["resident"] else "one launch per stage and round",
                "plan_taken_over": int(r.get("taken_over", 0)),
                "plan_clusters": int(r["clusters"]),
                "plan_us_per_round": 1e3 * r["ms_total"] / max(r["rounds"], 1)}
        # the host budget the plan ran under (frx_api.cpp host_cpu_share): CPUs of the process (cgroup quota / affinity), this rank's share, mailbox threads
        import ctypes as _C
        b_, s_, t_ = _C.c_double(), _C.c_int(), _C.c_int()
        frx.lib().frx_debug_host_cpu_share(int(r["clusters"]) or B, (world - 1) if lib_mode else 0, _C.byref(b_), _C.byref(s_), _C.byref(t_))
        plan.update({"plan_host_cpus": b_.value, "plan_host_cpu_share_of_this_rank": s_.value, "plan_mailbox_threads": t_.value, "local_world_size": os.environ.get("LOCAL_WORLD_SIZE")})
        try: plan["plan_mailbox_numa"] = prob.mailbox_numa()               # where the resident plan's mailbox pages live against the device's NUMA node (round 6: the "slow host" of round 5)
        except Exception as e_: plan["plan_mailbox_numa"] = {"error": repr(e_)}
        if world > 1 and not lib_mode:
            plan["plan_ms_per_rank"] = [v[0] for v in r["per_rank"]]
            plan["plan_us_per_round_per_rank"] = [1e3 * v[0] / max(v[1], 1.0) for v in r["per_rank"]]
            if r["one_device"]:
                plan["plan_us_per_round"] = plan["plan_us_per_round_per_rank"][0]
                plan["one_device_test"] = ("all ranks on ONE device (FRX_BENCH_DEVICE): the plans take turns (a resident grid needs the whole chip) while every waiting rank keeps one host "
                                           "thread spinning like its own mailbox thread would - the host side of the N-device job under this box's CPU quota; value and plan_ms are NOT measurements")
        if rank == 0:
            plan.update({"plan_ms_per_stage_path": r_ps["ms_total"], "plan_rounds_per_stage_path": r_ps["rounds"],
                         "plan_ms_one_candidate": r_b1["ms_total"], "plan_rounds_one_candidate": r_b1["rounds"],
                         "plan_us_per_round_one_candidate": 1e3 * r_b1["ms_total"] / max(r_b1["rounds"], 1),
                         "plan_setup_ms_one_candidate": t_setup1, "plan_initial_guess_ms_one_candidate": t_guess1,
                         "plan_ms_with_setup_one_candidate": r_b1["ms_total"] + t_setup1 + t_guess1,
                         "plan_path_one_candidate": "resident" if r_b1["resident"] else "per-stage",
                         "plans_per_s_per_stage_path": B / (r_ps["ms_total"] * 1e-3)})
            if r_q is not None:
                plan.update({"plan_ms_work_queue": r_q["ms_total"], "plan_clusters_work_queue": int(r_q["clusters"]), "plans_per_s_work_queue": B / (r_q["ms_total"] * 1e-3),
                             "plan_commands_of_the_busiest_cluster": r_q["rounds"], "plan_evals_mean": float(r_q["evals"].mean()),
                             "work_queue_equals_default_path_status": bool(np.array_equal(r_q["status"], r["status"])),
                             # the two paths form the direction differently (compact form in registers | two-loop recursion over HBM): same verdict per
                             # scenario is the claim, the L-BFGS code of a plan that stalls on an infeasible corridor may differ
                             "work_queue_status_mismatches": int((np.asarray(r_q["status"]) != np.asarray(r["status"])).sum()),
                             "work_queue_verdict_mismatches": int(((np.asarray(r_q["status"]) >= 0) != (np.asarray(r["status"]) >= 0)).sum())})
                # every scenario on which the two device paths disagree about success, with the CPU's verdict beside it: the four CPU variants of
                # tests/golden/mc512_cpu_verdicts.npz (ids 0 .. 511 = rank 0's share of configs[4]; data, not the oracle - the oracle is not called here)
                fix_path = os.path.join(ROOT, "tests", "golden", "mc512_cpu_verdicts.npz")
                fix = np.load(fix_path) if (args.config == "montecarlo4096" and os.path.exists(fix_path)) else None
                rows = []
                for b in range(B):
                    if (r_q["status"][b] >= 0) == (r["status"][b] >= 0): continue
                    row = {"scenario": rank * B + b, "default_path_status": int(r["status"][b]), "work_queue_status": int(r_q["status"][b]),
                           "default_path_objective": float(r["objective"][b]), "work_queue_objective": float(r_q["objective"][b])}
                    if fix is not None and rank * B + b < len(fix["status"]):
                        cs = [int(v) for v in fix["status"][rank * B + b]]
                        row.update({"cpu_status_four_variants": cs, "every_cpu_variant_fails_too": bool(max(cs) < 0)})
                    rows.append(row)
                plan["work_queue_verdict_mismatch_list"] = rows
                if fix is not None and rank == 0:
                    cs = fix["status"][:B]
                    inside = lambda st: int(sum(1 for b in range(B) if int(st[b]) in [int(v) for v in cs[b]] or cs[b].max() < 0))
                    plan.update({"plan_verdicts_inside_the_cpu_variants_set": inside(r["status"]), "work_queue_verdicts_inside_the_cpu_variants_set": inside(r_q["status"]),
                                 "cpu_verdict_source": "tests/golden/mc512_cpu_verdicts.npz: four CPU-oracle variants per scenario (tests/golden/make_mc_verdicts.py); a scenario every CPU variant fails on counts as inside whatever the device says"})
        if r_lib is not None:
            plan.update({"plan_front_end": "frx_multi_* (one process, one host thread + handle per device)", "plan_ms_whole_job": r_lib["ms_wall"],
                         "plan_shards": int(r_lib["n_shards"]), "plan_winner_exchange": r_lib["exchange"],
                         "plan_status_ok_whole_job": int(np.sum(r_lib["status"] >= 0)), "lib_winner_id": r_lib["winner_id"], "lib_winner_objective": r_lib["winner_objective"]})
            r["ms_total"] = r_lib["ms_wall"]                                 # the job's plan time: all shards, planned concurrently by the library
            plan["plan_ms"] = r_lib["ms_wall"]
        plan["plans_per_s"] = world * B / (r["ms_total"] * 1e-3)           # whole-job candidate optimisations per second
        if rank == 0:
            # The clock this box sustains under a latency-bound FP64 load (frx_debug_shader_clock: one lone wave per CU, a dependent FMA chain): a round is a chain of
            # dependent instructions, so round time x clock = cycles per round is the figure that should NOT move from box to box with one code object (VERDICT r5 weak 2:
            # the driver's box read 26.3 us per round where the builder's read 24.9)
            try:
                lo_, mean_, hi_ = frx.shader_clock(local_rank, 3.0)
                plan.update({"sclk_mhz_under_latency_bound_fp64_load": {"min": lo_, "mean": mean_, "max": hi_},
                             "plan_kilocycles_per_round": plan["plan_us_per_round"] * mean_ * 1e-3,
                             "plan_kilocycles_per_round_one_candidate": plan["plan_us_per_round_one_candidate"] * mean_ * 1e-3 if "plan_us_per_round_one_candidate" in plan else None})
            except Exception as e:
                plan["sclk_mhz_under_latency_bound_fp64_load"] = {"error": repr(e)}
        plan.update({"winner_id": gid, "winner_rank": owner, "winner_objective": obj, "winner_total_time_s": float(wT.sum())})
        if r_lib is not None:                                            # one process: the library selected the winner over all shards itself
            plan.update({"winner_id": r_lib["winner_id"], "winner_rank": int(r_lib["winner_id"] // B), "winner_objective": r_lib["winner_objective"], "winner_total_time_s": float(r_lib["winner_T"].sum())})

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cands, params, kappa, x_state, prob.x_off)
    boundary = None
    if rank == 0:
        try:
            boundary = boundary_calls(frx, sc, prob, cands, params, kappa, x_state, local_rank, with_cpu=not args.no_cpu_baseline)
        except Exception as e:                                              # a failing side leg must not cost the bench line
            boundary = {"error": repr(e)}

    if rank == 0 and cpu is not None and plan:
        # The 1e-6 contract on optimised coefficients, where the number is: the device's plans against the CPU oracle's plans of
        # the same candidates, next to the CPU oracle against itself with one rounding moved (the sample abscissa's other form) - independent runs of the
        # reference's stop rule end this far apart; the lock-step form (tests/test_gpu_parity.py) is where 1e-9 ... 1e-13 holds
        def spread(pa, pb):
            worst_c, worst_f = 0.0, 0.0
            for b, (qa, qb) in enumerate(zip(pa, pb)):
                worst_c = max(worst_c, float(np.abs(qa["C"] - qb["C"]).max() / np.abs(qb["C"]).max()))
                worst_f = max(worst_f, abs(float(qa["objective"]) - float(qb["objective"])) / abs(float(qb["objective"])))
            return worst_c, worst_f
        nb = len(cpu_baseline.plans)
        dev_plans = [{"C": r["C"][6 * prob.piece_off[b]:6 * prob.piece_off[b + 1]], "objective": r["objective"][b]} for b in range(nb)]
        sc_dev, sf_dev = spread(dev_plans, cpu_baseline.plans)
        sc_cpu, sf_cpu = spread(cpu_baseline.plans_other_rounding, cpu_baseline.plans)
        plan.update({"plan_coeff_spread_vs_cpu": sc_dev, "plan_objective_spread_vs_cpu": sf_dev, "plan_coeff_spread_cpu_vs_cpu": sc_cpu, "plan_objective_spread_cpu_vs_cpu": sf_cpu,
                     "plan_spread_definition": f"max over the first {nb} candidates of max|C_a - C_b| / max|C_b| (and |f_a - f_b| / |f_b|) between INDEPENDENT optimiser runs at the stock "
                                               "tolerance; cpu_vs_cpu = the CPU oracle against itself with the sample abscissa accumulated (CPU.hpp:400) instead of multiplied (cc.cu:152)"})

    # The kernel that actually runs a plan (VERDICT r3 missing 3): k_round, priced per ROUND = one evaluation + one L-BFGS update of every candidate.
    # Bytes: SURVEY.md 8d's full-evaluation bytes (the history never touches HBM); FP64 work: the penalty integrator's counted flops plus the direction
    # (pass A 4 m n FMAs, dense 3 m^2, pass B 2 m n per accepted step).  Idle fractions from the committed round budget (instrumented instantiation).
    round_obj = None
    if rank == 0 and plan and fp64:
        m_hist = 128
        us_round = plan["plan_us_per_round"]
        dir_flops = float(np.sum(2.0 * (6 * m_hist * nx + 3 * m_hist * m_hist)))
        pen_flops = fp64["flops_per_sample"] * samples_per_step
        round_obj = {"kernel": "frx::k_round (resident: one launch per plan)", "us_per_round": us_round, "bytes_per_round": eval_bytes,
                     "achieved": eval_bytes / (us_round * 1e-6) / 1e9, "unit": "GB/s", "frac": eval_bytes / (us_round * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "fp64_flops_per_round": pen_flops + dir_flops, "fp64_flops_penalty": pen_flops, "fp64_flops_direction": dir_flops,
                     "fp64_frac": (pen_flops + dir_flops) / (us_round * 1e-6) / 1e12 / FP64_PEAK_TFLOPS,
                     "bound": "latency: per candidate a round is ONE dependent chain (direction -> forward map -> penalty -> adjoint) on 8 of the chip's 256 CUs"}
        for name in ("r06_round_budget_B32.json", "r05_round_budget_B32.json", "r04_round_budget_B32.json", "r03_round_budget_B32.json"):
            bp = os.path.join(ROOT, "profiles", name)
            if os.path.exists(bp) and args.config == "headline":
                try:
                    bj = json.loads(open(bp).read().split("\n{\"per_stage")[0])
                    tot = bj["leader"]["total"]
                    round_obj["budget"] = {"from_profile": True, "source": "profiles/" + name, "instrumented_us_per_round": bj["us_per_round_wall"],
                                           "leader_us": bj["leader"], "leader_keys": bj.get("what the keys mean for the leader"),
                                           # (since the direction and the penalty partials arrive as granules the leader's waits for the cluster are inside `gather` and `backward`:
                                           # idle = those two minus the gather's own ~1.0 us and the adjoint's ~5.4 us is not separable here - scripts/r04/round_gaps.py has the timeline)
                                           "leader_idle_frac": (bj["leader"].get("wait_arrive", 0.0) + bj["leader"].get("wait_host", 0.0)) / tot,
                                           "member_idle_frac": (bj["member1"].get("wait_phase", 0.0) + bj["member1"].get("wait_u", 0.0)) / bj["member1"]["total"],
                                           "dense_idle_frac": (bj["dense"].get("wait_phase", 0.0) + bj["dense"].get("wait_part", 0.0)) / bj["dense"]["total"]}
                except Exception as e:                                   # a malformed profile must not cost the bench line
                    round_obj["budget"] = {"error": repr(e), "source": "profiles/" + name}
                break

    if rank == 0:
        out = {
            "metric": "constraint-samples/s", "value": world * samples_per_step * args.steps / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_host_wall": dt_wall / args.steps * 1e3, "value_host_wall": world * samples_per_step * args.steps / dt_wall,
            "timing": "HIP events on the launch stream around the K steps (first launch -> last kernel end), inside the barrier + synchronize bracket whose host wall clock is ms_per_step_host_wall; max over ranks",
            "launch": ((f"ONE hipGraph of the K steps ({'K kernel nodes: one k_eval_cluster launch per step' if fused_G else 'K kernel nodes: one k_eval_solo launch per step' if solo_wgs else '3 K kernel nodes'}, captured from K calls of frx_objective_eval_device)") if graph is not None
                       else (graph_note or ("K direct launches of k_eval_cluster" if fused_G else "K direct launches of k_eval_solo" if solo_wgs else "K x 3 direct kernel launches"))),
            "evaluation_form": ({"kernel": "frx::k_eval_cluster", "workgroups_per_candidate": fused_G, "what": "one launch per evaluation: a cluster of workgroups per candidate (leader: forward map and adjoint; members: penalty integral), csrc/frx_eval_kernel.hpp",
                                 "us_per_evaluation_back_to_back_launches": one_us, "us_per_evaluation_as_three_stage_launches": three_us}
                                if fused_G else
                                {"kernel": "frx::k_eval_solo", "workgroups_per_cu": solo_wgs, "what": "one launch per evaluation: ONE workgroup per candidate runs forward map, penalty integral of its own pieces and adjoint back to back (the stage kernels' bodies; bit-identical results), csrc/frx_solo_kernel.hpp",
                                 "us_per_evaluation_back_to_back_launches": solo_us, "us_per_evaluation_as_three_stage_launches": three_us}
                                if solo_wgs else {"kernel": "three stage kernels", "us_per_evaluation_back_to_back_launches": three_us}),
            "warmup_graph_replays": GRAPH_WARM_REPLAYS if graph is not None else 0,
            "ms_per_step_graph_replayed_back_to_back": (dt_b2b / args.steps * 1e3) if dt_b2b else None,
            "ms_per_step_direct_launches": dt_direct / args.steps * 1e3, "value_direct_launches": world * samples_per_step * args.steps / dt_direct,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {B} candidate trajs/GPU x {N} pieces x {kappa} quadrature intervals "
                                   f"({samples_per_step} constraint samples/step/GPU), 16-gate Zhangjiajie-like corridor, K_i=8",
                       "step": "one batched objective evaluation x->(f,grad), inputs resident in HBM: " + ("forward map + penalty integral + adjoint in ONE launch (k_eval_cluster)" if fused_G else "forward map + penalty integral + adjoint in ONE launch (k_eval_solo: one workgroup per candidate)" if solo_wgs else "k_forward + k_penalty + k_backward"),
                       "state": "iterate after 60 L-BFGS iterations from the reference initial guess",
                       "parallelism": f"candidates sharded {B}/GPU, no data-path collective",
                       "front_end": ("frx_multi_* in one process" if lib_mode else "one process per GPU (torch.distributed)") if world > 1 else "one process, one device"},
            "roofline": {**({"bound": "hbm", "kernel": "frx::k_eval_cluster" if fused_G else "frx::k_eval_solo", "selected_by": "the dominant (only) kernel of the timed region: one launch per evaluation",
                             "achieved": eval_bytes / (one1_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": eval_bytes / (one1_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": eval_bytes, "algorithmic_bytes_definition": "SURVEY.md 8d, full objective evaluation fused on device: penalty bytes (sum over pieces of 312 + 48 K_i) + 16 n + 24 sum(nv) per candidate",
                             "avg_kernel_us": one1_us, "measured_by": "HIP events on the library's launch stream around back-to-back launches of this kernel, in this run (frx_eval_launch_time)",
                             "traffic": (0.5 * (one_traffic[0] + one_traffic[1]) if one_traffic else None), "traffic_range": one_traffic, "traffic_from_profile": one_traffic is not None, "traffic_source": traffic_src if one_traffic else None,
                             "traffic_note": "FETCH_SIZE / WRITE_SIZE of the kernel, bracketed by the 8-byte and the 16-byte calibration of the same call (its loads are a mix); includes the granules that carry (C, T) and the partials between the workgroups (2 x 16 bytes per value)",
                             "counter_calibration_bytes_per_counted_byte": calib,
                             "kernel_samples_per_s": samples_per_step / (one1_us * 1e-6), "workgroups_per_candidate": fused_G if fused_G else 1,
                             "bound_in_fact": ("latency: per candidate one dependent chain - forward map (one wave per axis behind a matrix wave), penalty share of 24 member waves, adjoint - on 7 of 256 CUs; nothing of it streams" if fused_G else
                                               "latency: per candidate one dependent chain in ONE workgroup - forward map, five penalty passes on its four waves, adjoint - two workgroups per CU (DESIGN.md 3.8)"),
                             "penalty_integrator": {"what": "the penalty integrator alone, as a stage kernel (the kernel SURVEY.md 8d prices; it is what large batches and the optimiser's per-stage rounds launch)",
                                                    "kernel": pen_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                                    "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_definition": "sum over pieces of 312 + 48 K_i (SURVEY.md 8d)", "avg_kernel_us": pen_us,
                                                    "measured_by": "HIP events on the library's launch stream around back-to-back launches of this kernel, in this run (frx_eval_stage_times)",
                                                    "traffic": traffic, "traffic_from_profile": traffic is not None, "traffic_source": traffic_src,
                                                    "kernel_samples_per_s": samples_per_step / (pen_us * 1e-6), "sample_halfspace_pairs_per_s": pairs_per_step / (pen_us * 1e-6)}}
                            if (fused_G or solo_wgs) else
                            {"bound": "hbm", "kernel": pen_kernel, "selected_by": "the kernel SURVEY.md 8d prices: the penalty integrator (CPU.hpp:188-408 = cuda_computer::compute)",
                             "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_definition": "sum over pieces of 312 + 48 K_i (SURVEY.md 8d)", "avg_kernel_us": pen_us,
                             "measured_by": "HIP events on the library's launch stream around back-to-back launches of this kernel, in this run (frx_eval_stage_times)",
                             "traffic": traffic, "traffic_from_profile": traffic is not None, "traffic_source": traffic_src, "counter_calibration_bytes_per_counted_byte": calib,
                             "kernel_samples_per_s": samples_per_step / (pen_us * 1e-6), "sample_halfspace_pairs_per_s": pairs_per_step / (pen_us * 1e-6)}),
                         "fp64": fp64, "valu": valu, "large_batch": large,
                         "evaluation": {"what": "one whole step x -> (f, grad) as timed (`value`): forward map + penalty integrator + adjoint, " + ("one launch (clusters)" if fused_G else "one launch (one workgroup per candidate)" if solo_wgs else "three launches"),
                                        "algorithmic_bytes_per_step": eval_bytes, "definition": "SURVEY.md 8d: penalty bytes + 16 n + 24 sum(nv) per candidate",
                                        "us_per_step": dt / args.steps * 1e6, "achieved": eval_bytes / (dt / args.steps) / 1e9, "unit": "GB/s",
                                        "frac": eval_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
                         "stage_kernels_us": stage_us,
                         "stage_kernels_note": "the three stage kernels of an evaluation, one at a time" + (" (not what the timed steps launched: see evaluation_form)" if (fused_G or solo_wgs) else ""),
                         "knot_kernels": {k: {"kernel": {"forward": "frx::k_forward_knot64", "adjoint": "frx::k_backward_knot64"}[k] if N <= 64 else {"forward": "frx::k_forward_knot", "adjoint": "frx::k_backward_knot"}[k], "avg_kernel_us": stage_us[k],
                                              "implementation_traffic_bytes": stage_bytes[k],
                                              "implementation_traffic_note": "x, waypoint polytopes, (T, C), out20, saved reduction multipliers, g: stage buffers between the three launches, NOT algorithmic bytes",
                                              "implementation_traffic_frac_of_hbm_peak": stage_bytes[k] / (stage_us[k] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                              "traffic_range": knot_traffic.get(k), "traffic_from_profile": knot_traffic.get(k) is not None,
                                              "bound": "latency: one workgroup per candidate, a dependent FP64 chain (DESIGN.md 3.1, 3.3)"} for k in ("forward", "adjoint")},
                         "round": round_obj, "hbm_bound_kernel": hbm_kernel, "states": states},
            "cpu_baseline": cpu,
            "boundary_call_us": boundary,
        }
        out.update(plan)
        print(json.dumps(out), flush=True)
    prob.close()
    for sh in lib_shards: sh[0].close()


if __name__ == "__main__":
    main()
