"""Plan time with the HOST busy: the headline plan (32 candidates, and one) while N other processes spin on the box's cores.
   python scripts/r04/busy_host.py [spinners ...]      e.g. 0 64 192 256"""
import json, multiprocessing as mp, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def spin(stop):
    x = 1.0
    while not stop.value:
        for _ in range(100000): x = x * 1.0000001 + 1e-9

def measure():
    from frx_import import frx
    from fast_racing_amd import scenario as sc
    out = {}
    for B in (32, 1):
        cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)]
        prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
        x0 = prob.initial_guess(); tol = sc.ZHANGJIAJIE["opt_rel_tol"]
        prob.optimize(tol, x0=x0, max_iterations=50)
        rs = [prob.optimize(tol, x0=x0) for _ in range(3)]
        out[f"B={B}"] = {"us_per_round": [round(1e3 * r["ms_total"] / r["rounds"], 2) for r in rs], "plan_ms": [round(r["ms_total"], 1) for r in rs]}
        prob.close()
    return out

if __name__ == "__main__":
    counts = [int(a) for a in sys.argv[1:]] or [0, 64, 256]
    res = {"cpus": os.cpu_count(), "loadavg_before": os.getloadavg()[0]}
    for n in counts:
        stop = mp.Value("i", 0)
        ps = [mp.Process(target=spin, args=(stop,), daemon=True) for _ in range(n)]
        for p in ps: p.start()
        time.sleep(1.0 if n else 0.0)
        res[f"{n} spinning processes"] = measure()
        stop.value = 1
        for p in ps: p.join(timeout=5)
    print(json.dumps(res))
