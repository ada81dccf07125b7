"""Multi-GPU plumbing: candidates are independent optimisation problems, so the
batch is block-partitioned over ranks and an evaluation needs NO collective.  The only exchange
of a plan is the final winner selection: all-gather of (objective, global candidate id) — 16 B per
rank — then a broadcast of the winner's (6N x 3) coefficients and N durations from its owner.
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int):
    """Block partition [lo, hi) of `total` candidates for `rank`; earlier ranks take the remainder."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def select_winner(dist, device, local_objective: np.ndarray, local_ids: np.ndarray, local_coeffs, local_T, n_pieces_max: int,
                  local_status=None):
    """All ranks learn the best candidate of the whole job.

    local_coeffs(i) / local_T(i) return the coefficient block (6N x 3) / durations (N) of local
    candidate i.  Returns (global_id, objective, owner_rank, coeffs, T).  Ties go to the lowest
    global id so every rank takes the same decision.  Candidates whose optimiser failed (`local_status` < 0: the line search
    gave up and the point was reverted) or whose objective is not finite never win: their objective counts as +inf."""
    import torch
    local_objective = np.array(local_objective, dtype=np.float64, copy=True)
    bad = ~np.isfinite(local_objective)
    if local_status is not None:
        bad |= np.asarray(local_status) < 0
    local_objective[bad] = np.inf
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    if len(local_objective):
        order = np.lexsort((local_ids, local_objective))
        best = int(order[0])
        mine = torch.tensor([float(local_objective[best]), float(local_ids[best])], dtype=torch.float64, device=device)
    else:
        best = -1
        mine = torch.tensor([float("inf"), -1.0], dtype=torch.float64, device=device)
    if world > 1:
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        table = torch.stack(allv).cpu().numpy()
    else:
        table = mine.cpu().numpy()[None, :]
    order = np.lexsort((table[:, 1], table[:, 0]))
    owner = int(order[0])
    gid, obj = int(table[owner, 1]), float(table[owner, 0])
    payload = torch.zeros(1 + n_pieces_max * 19, dtype=torch.float64, device=device)
    if rank == owner:
        c = np.asarray(local_coeffs(best), dtype=np.float64).reshape(-1)
        t = np.asarray(local_T(best), dtype=np.float64).reshape(-1)
        n = t.size
        buf = np.zeros(1 + n_pieces_max * 19)
        buf[0] = n; buf[1:1 + 18 * n] = c; buf[1 + 18 * n_pieces_max:1 + 18 * n_pieces_max + n] = t
        payload.copy_(torch.from_numpy(buf))
    if world > 1:
        dist.broadcast(payload, src=owner)
    buf = payload.cpu().numpy()
    n = int(buf[0])
    coeffs = buf[1:1 + 18 * n].reshape(6 * n, 3).copy()
    T = buf[1 + 18 * n_pieces_max:1 + 18 * n_pieces_max + n].copy()
    return gid, obj, owner, coeffs, T


# ---- self-launch of the one-process-per-GPU front end (bench.py --gpus N without an external launcher) ----
def self_launch_command(n_ranks: int, script: str, argv, port: int = 0):
    """The command that starts `script argv` as n_ranks ranks of one node.  With a port: exactly what the driver runs for N > 1
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script ...`).  Without one
    (what self_launch uses): `--standalone --local-addr 127.0.0.1`, whose rendezvous store binds port 0 itself - a port found free by a probe
    socket here could be taken again before the launcher binds it."""
    import sys
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}"]
    if port:
        return head + ["--master-addr", "127.0.0.1", "--master-port", str(int(port)), script] + list(argv)
    return head + ["--standalone", "--local-addr", "127.0.0.1", script] + list(argv)


def ranks_to_launch(n_gpus: int, env, visible_devices: int) -> int:
    """How many ranks THIS process has to start: 0 when it already is a rank of a launched job (WORLD_SIZE set) or when one rank was asked for.
    Fails loudly - never degrades - when the job the flags describe cannot be what runs:
      * a launcher started a world whose size is not --gpus;
      * fewer devices are visible than ranks asked for (unless FRX_BENCH_DEVICE pins every rank to one device: the 1-GPU-box test knob)."""
    n_gpus = int(n_gpus)
    if n_gpus < 1:
        raise SystemExit(f"--gpus {n_gpus}: at least one rank")
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != n_gpus:
            raise SystemExit(f"--gpus {n_gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a {world}-rank run as {n_gpus} GPUs")
        need = 0
    else:
        need = n_gpus if n_gpus > 1 else 0
    if n_gpus > 1 and not env.get("FRX_BENCH_DEVICE") and visible_devices < n_gpus:
        raise SystemExit(f"--gpus {n_gpus} needs {n_gpus} HIP devices, {visible_devices} visible (FRX_BENCH_DEVICE=<d> puts every rank on one device: a control-flow test, not a measurement)")
    return need


def self_launch(n_ranks: int, script: str, argv, env=None) -> int:
    """Run `script argv` as n_ranks ranks (one process per GPU) and return the job's exit code; rank 0's stdout is this process's stdout."""
    import os
    import subprocess
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on these hosts
    e.setdefault("OMP_NUM_THREADS", "1")
    e["FRX_BENCH_SELF_LAUNCHED"] = "1"                       # the ranks know who started them (any launcher sets TORCHELASTIC_RUN_ID)
    return subprocess.call(self_launch_command(n_ranks, script, argv), env=e)
