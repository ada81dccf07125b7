"""How long does the single-process RCCL communicator of frx_multi_create take to come up, per environment? (one fresh process per variant)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import os, sys, time
sys.path.insert(0, %r)
from frx_import import frx
from fast_racing_amd import scenario as sc
cands = [sc.make_candidate(0, 16, 4, perturb_id=b) for b in range(2)]
t0 = time.perf_counter()
mp = frx.MultiProblem(cands, sc.ZHANGJIAJIE, qd_intervals=8)
t1 = time.perf_counter()
r = mp.optimize(1e-5, max_iterations=20)
t2 = time.perf_counter()
print("create %%.2fs optimize+exchange %%.2fs rccl %%s exchange %%s" %% (t1 - t0, t2 - t1, mp.uses_rccl, r["exchange"]))
mp.close()
''' % ROOT
variants = {"default": {}, "ib_off_lo": {"NCCL_IB_DISABLE": "1", "NCCL_SOCKET_IFNAME": "lo"}, "net_socket": {"NCCL_NET": "Socket", "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1"},
            "debug": {"NCCL_DEBUG": "INFO"}}
for name, env in variants.items():
    e = dict(os.environ); e.update(env)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", child], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400)
    out = r.stdout.strip().splitlines()
    keep = [l for l in out if l.startswith("create")]
    print(name, "wall %.1fs" % (time.perf_counter() - t0), keep[-1] if keep else out[-3:])
    if name == "debug":
        open(os.path.join(ROOT, "gpurun_out", "rccl_init_debug.txt"), "w").write(r.stdout)
