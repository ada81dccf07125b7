"""Multi-rank logic on CPU: world_size 2, gloo.  The data path has no collective (candidates are independent);
what is exercised here is the sharding arithmetic and the winner selection exchange used after a plan."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, B, N, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from frx_import import frx  # noqa: F401
    from fast_racing_amd.dist import select_winner, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(B, rank, world)
    rng = np.random.default_rng(1234)                       # same table on every rank: the "global truth"
    cost = rng.uniform(1e4, 2e4, B); cost[5] = cost.min() - 1.0   # unique winner = candidate 5
    coeffs = rng.standard_normal((B, 6 * N, 3)); Ts = rng.uniform(0.2, 0.5, (B, N))
    ids = np.arange(lo, hi)
    gid, obj, owner, wc, wT = select_winner(dist, torch.device("cpu"), cost[lo:hi], ids, lambda i: coeffs[lo + i], lambda i: Ts[lo + i], N)
    ok = gid == 5 and abs(obj - cost[5]) == 0 and np.array_equal(wc, coeffs[5]) and np.array_equal(wT, Ts[5]) and lo <= 5 < hi if rank == owner else gid == 5
    ok = ok and np.array_equal(wc, coeffs[5]) and np.array_equal(wT, Ts[5])
    q.put((rank, bool(ok), (lo, hi), owner))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything_once(frx):
    from fast_racing_amd.dist import shard_range
    for total in (1, 7, 32, 256, 4096):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_winner_selection_two_ranks_gloo(frx):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    B, N = 12, 4
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, N, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok, _, _ in res), res
    assert {r for r, *_ in res} == {0, 1}
    assert all(owner == 0 for *_, owner in res)             # candidate 5 lives on rank 0 of 2 (6 candidates each)


def test_single_rank_winner(frx):
    import torch
    from fast_racing_amd.dist import select_winner
    cost = np.array([3.0, 1.0, 2.0]); ids = np.array([10, 11, 12])
    gid, obj, owner, wc, wT = select_winner(None, torch.device("cpu"), cost, ids, lambda i: np.full((12, 3), float(i)), lambda i: np.full(2, 0.5), 2)
    assert (gid, obj, owner) == (11, 1.0, 0) and np.all(wc == 1.0) and wT.sum() == 1.0


# ---- bench.py --gpus N starts its own ranks: the launch path, on CPU ----
def _bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FRX_BENCH_DEVICE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_gpus_n_launches_n_ranks_itself(frx):
    import json
    p = _bench(["--gpus", "2", "--launch-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["launch_check"] is True and line["n_gpus"] == 2 and line["self_launched"] is True
    # one rank: no launcher involved at all
    p = _bench(["--gpus", "1", "--launch-check"])
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_refuses_to_degrade(frx):
    # fewer devices than ranks (this container has none, a 1-GPU box has one): an error that says so, not a 1-rank run labelled N GPUs
    if frx.lib().frx_device_count() < 2:
        p = _bench(["--gpus", "2", "--no-plan", "--no-cpu-baseline"])
        assert p.returncode != 0 and "needs 2 HIP devices" in p.stderr
        p = _bench(["--gpus", "2", "--multi", "lib", "--no-plan", "--no-cpu-baseline"])
        assert p.returncode != 0 and "needs 2 HIP devices" in p.stderr
    # a launcher whose world is not --gpus
    p = _bench(["--gpus", "2", "--launch-check"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr


def test_self_launch_command_is_the_drivers(frx):
    from fast_racing_amd.dist import self_launch_command, ranks_to_launch
    cmd = self_launch_command(4, "/x/bench.py", ["--gpus", "4", "--steps", "5"], port=29400)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29400"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "5"]
    own = self_launch_command(2, "/x/bench.py", ["--gpus", "2"])              # bench.py's own launch: the rendezvous store picks its port itself
    assert "--standalone" in own and own[own.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in own and own[-3:] == ["/x/bench.py", "--gpus", "2"]
    assert ranks_to_launch(1, {}, 1) == 0 and ranks_to_launch(8, {}, 8) == 8 and ranks_to_launch(8, {"WORLD_SIZE": "8"}, 8) == 0
    assert ranks_to_launch(2, {"FRX_BENCH_DEVICE": "0"}, 1) == 2          # the 1-GPU-box knob: every rank on one device
    with pytest.raises(SystemExit):
        ranks_to_launch(2, {}, 1)
