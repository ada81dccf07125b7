"""FP64 work of one constraint sample of k_penalty, counted in the gfx950 ISA hipcc emits (an FP64 roofline figure next to the HBM one).
Static count per region of the kernel's control flow, from the compiler's loop annotations of a build whose half-space loops are not unrolled
(-DFRX_COUNT_BUILD, frx_math.hpp):
  fixed      one pass of the sample loop outside the half-space chunk loop (attitude, limits, reverse passes, the beta (x) a products)
  hs_pre     round 5: the pre-reject of ONE half-space (signed distance of the centre against -max(ell)) - every sample executes it
  hs_test    the full sign test of one half-space - executed for a chunk of 4 only when some lane of the wave is within max(ell) of one of them
  hs_viol    the extra work of a VIOLATED half-space (sqrt, cube, gradient accumulation)
flops: v_fma_f64 / v_fmac_f64 = 2, every other FP64 VALU instruction (add, mul, rcp, rsq, sqrt, div_*, min/max, ldexp, ...) = 1.
The DYNAMIC count of a real launch (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 per wave, scripts/r05/gpu_pmc.sh) is what bench.py quotes when a
counter pass of the same call is there; this static count is the CPU-side cross-check.
Usage: python scripts/count_fp64.py [K] [thr]   -> JSON"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
src = os.path.join(ROOT, "fast-racing_amd", "csrc", "frx_device.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "frx.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DFRX_COUNT_BUILD", "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
sym = "_ZN3frx13k_penalty_latE" if (len(sys.argv) <= 2 or sys.argv[2] != "thr") else "_ZN3frx9k_penaltyE"   # default form of the kernel: latency form
start = next(i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().endswith(":") or l.startswith(sym) and ": ; @" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
f64 = re.compile(r"^\s+(v_[a-z0-9_]*_f64)(?:_e32|_e64|_dpp|_sdwa)?\b")
def flops_of(l):
    m = f64.match(l)
    if not m or m.group(1).startswith(("v_cmp", "v_cvt", "v_mov", "v_cndmask", "v_readlane")): return 0, 0
    return 1, (2 if "fma" in m.group(1) else 1)
# basic blocks with the innermost loop they belong to: ".LBBx_y:   ; in Loop: Header=BBx_z Depth=d" / "=>This ... Header: Depth=d"
blocks, cur = [], None
for i, l in enumerate(body):
    m = re.match(r"^\.L(BB\d+_\d+):\s*(?:;\s*(.*))?$", l)
    if m or re.match(r"^; %bb\.\d+:\s*(?:;\s*(.*))?$", l):
        note = (m.group(2) if m else re.match(r"^; %bb\.\d+:\s*(?:;\s*(.*))?$", l).group(1)) or ""
        name = m.group(1) if m else None
        j = i + 1
        while j < len(body) and body[j].strip().startswith(";") and "Loop" in body[j]:
            note += " " + body[j].strip(); j += 1
        hdr = depth = None
        mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
        if mm: hdr, depth = mm.group(1), int(mm.group(2))
        mm = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", note)
        if mm: hdr, depth = name, int(mm.group(1))
        cur = {"hdr": hdr, "depth": depth, "n": 0, "fl": 0}
        blocks.append(cur)
    elif cur is not None:
        n, fl = flops_of(l)
        cur["n"] += n; cur["fl"] += fl
loops = {}
for b in blocks:
    if b["hdr"] is None: continue
    e = loops.setdefault(b["hdr"], {"depth": b["depth"], "n": 0, "fl": 0})
    e["n"] += b["n"]; e["fl"] += b["fl"]
# the sample loop is the depth-1 loop that has children (the staging loops have none); its depth-2 child is the chunk loop, whose depth-3 children
# are, in source order, the pre-reject loop, the full-test loop and the slow pass
order = [h for h in dict.fromkeys(b["hdr"] for b in blocks if b["hdr"])]
d3 = [h for h in order if loops[h]["depth"] == 3]
d2 = [h for h in order if loops[h]["depth"] == 2]
first_d3 = min(order.index(h) for h in d3)
chunk = max((h for h in d2 if order.index(h) < first_d3), key=order.index)
sample = max((h for h in order if loops[h]["depth"] == 1 and order.index(h) < order.index(chunk)), key=order.index)
tot = lambda hs: (sum(loops[h]["n"] for h in hs), sum(loops[h]["fl"] for h in hs))
inner = [h for h in order if order.index(h) >= order.index(chunk) and loops[h]["depth"] >= 2 and (h == chunk or h in d3 and order.index(h) < order.index(chunk) + 1 + 3)]
n_in, f_in = tot(inner)
n_s, f_s = loops[sample]["n"] + n_in, loops[sample]["fl"] + f_in
d3w = [h for h in d3 if loops[h]["n"] > 0]                      # (the loop that only loads the chunk's records has no FP64 instruction)
assert len(d3w) >= 3, ("expected pre-reject, full-test and slow-pass loops at depth 3", d3w)
pre, full, viol = (loops[h] for h in d3w[:3])
res = {"kernel": "frx::k_penalty_lat (default form)" if "lat" in sym else "frx::k_penalty (throughput form, FRX_PENALTY_FORM=thr)", "K": K,
       "note": "static count of a build whose half-space loops are not unrolled; `fixed` includes the 20 adds of the accumulate path that only runs with more than one sample per lane (kappa + 1 > 64) "
               "and the conditional reverse passes of violated limits (statically present, dynamically skipped by a wave without violations); the chunk loop's own few instructions are counted as fixed",
       "fp64_instructions": {"fixed": loops[sample]["n"] + loops[chunk]["n"], "hs_pre": pre["n"], "hs_test": full["n"], "hs_violated_extra": viol["n"]},
       "flops": {"fixed": loops[sample]["fl"] + loops[chunk]["fl"], "hs_pre": pre["fl"], "hs_test": full["fl"], "hs_violated_extra": viol["fl"]}}
fx = res["flops"]
res["flops_per_sample_no_violation"] = fx["fixed"] + K * fx["hs_pre"]                                       # every half-space pre-rejected (98-99 % of the pairs on the headline batch)
res["flops_per_sample_no_violation_all_tested"] = fx["fixed"] + K * (fx["hs_pre"] + fx["hs_test"])           # the round-4 work: no pre-reject succeeds
res["flops_per_sample_all_violated"] = fx["fixed"] + K * (fx["hs_pre"] + fx["hs_test"] + fx["hs_violated_extra"])
print(json.dumps(res))
