"""FP64 work of one constraint sample of k_penalty, counted in the gfx950 ISA hipcc emits (an FP64 roofline figure next
to the HBM one).  Static count per region of the kernel's control flow:
  fixed     one pass of the sample loop outside the half-space loop (attitude, limits, reverse passes, 6x3 outer products)
  hs_test   the part of a half-space iteration every sample executes (distance + sign test)
  hs_viol   the extra work of a VIOLATED half-space (sqrt, cube, gradient accumulation)
flops: v_fma_f64 = 2, every other FP64 VALU instruction (add, mul, rcp, rsq, sqrt, div_*, min/max, ldexp, frexp, trig_preop) = 1.
Usage: python scripts/count_fp64.py [K]   -> JSON with flops per sample for K half-spaces, none / all of them violated."""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
src = os.path.join(ROOT, "fast-racing_amd", "csrc", "frx_device.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "frx.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DFRX_COUNT_BUILD", "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)   # FRX_COUNT_BUILD: the half-space loop is not unrolled (frx_math.hpp)
    lines = open(out).read().split("\n")
sym = "_ZN3frx13k_penalty_latE" if (len(sys.argv) <= 2 or sys.argv[2] != "thr") else "_ZN3frx9k_penaltyE"   # default form of the kernel: latency form
start = next(i for i, l in enumerate(lines) if l.startswith(sym))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
f64 = re.compile(r"^\s+(v_[a-z0-9_]*_f64)(?:_e32|_e64|_dpp|_sdwa)?\b")
def flops(seg):
    n = fl = 0
    for l in seg:
        m = f64.match(l)
        if m and not m.group(1).startswith(("v_cmp", "v_cvt", "v_mov", "v_cndmask", "v_readlane")):
            n += 1; fl += 2 if "fma" in m.group(1) else 1
    return n, fl
# regions by loop annotations of the compiler
hdr1 = next(i for i, l in enumerate(body) if "=>This Loop Header: Depth=1" in l)                      # sample loop
name1 = re.match(r"\.L(BB\d+_\d+):", body[hdr1 - 1] if body[hdr1].strip().startswith(";") else body[hdr1]).group(1)
in1 = [i for i, l in enumerate(body) if f"Header={name1} Depth=1" in l]
tail1 = next(i for i in range(max(in1) + 1, len(body)) if body[i].startswith(".LBB"))
hs_hdr = next(i for i, l in enumerate(body) if f"Parent Loop {name1} Depth=1" in l)                    # half-space loop header (depth 2)
name2 = re.match(r"\.L(BB\d+_\d+):", body[hs_hdr]).group(1)
in2 = [i for i, l in enumerate(body) if f"Header={name2} Depth=2" in l] + [hs_hdr]
hs_lo, hs_hi = min(in2), next(i for i in range(max(in2) + 1, len(body)) if body[i].startswith(".LBB"))
first_branch = next(i for i in range(hs_hdr, hs_hi) if "s_cbranch" in body[i] or "s_branch" in body[i])
sample = body[hdr1:tail1]
n_all, f_all = flops(sample)
n_hs, f_hs = flops(body[hs_lo:hs_hi])
n_test, f_test = flops(body[hs_hdr:first_branch])
res = {"kernel": "frx::k_penalty_lat (default form)" if "lat" in sym else "frx::k_penalty (throughput form, FRX_PENALTY_FORM=thr)", "K": K,
       "note": "static count of a build whose half-space loop is not unrolled; includes the 20 adds of the accumulate path that only runs with more than one sample per lane (kappa + 1 > 64)",
       "fp64_instructions": {"fixed": n_all - n_hs, "hs_test": n_test, "hs_violated_extra": n_hs - n_test},
       "flops": {"fixed": f_all - f_hs, "hs_test": f_test, "hs_violated_extra": f_hs - f_test}}
res["flops_per_sample_no_violation"] = res["flops"]["fixed"] + K * f_test
res["flops_per_sample_all_violated"] = res["flops"]["fixed"] + K * f_hs
print(json.dumps(res))
