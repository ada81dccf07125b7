# HBM traffic of k_lbfgs_pre at 256 candidates (the HBM-bound kernel of the path): FETCH_SIZE / WRITE_SIZE in separate passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/dv256.py <<PY
import sys; sys.path.insert(0, "$R")
import frx_import, fast_racing_amd as frx
print(frx.dv_selftest(641, B=256, m=128, iters=160))
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcl_$c -o p -- python /tmp/dv256.py > /dev/null 2> $R/gpurun_out/pmcl_$c.err
done
cd $R
python - <<'PY'
import csv, json
out = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for r in csv.DictReader(open(f"gpurun_out/pmcl_{name}/p_counter_collection.csv")) if "k_lbfgs_pre" in r["Kernel_Name"]]
    v = sorted(float(r["Counter_Value"]) for r in rows)[-20:]          # the launches with a full history (bound = 128)
    out[name] = {"launches": len(v), "mean_kb": sum(v) / len(v)}
row = 768; alg = 256 * 128 * 2 * 2 * row * 8
res = {"command": "frx_dv_selftest(n=641, B=256, m=128, iters=160) [scripts/gpu_pmc_lbfgs.sh]", "kernel": "frx::k_lbfgs_pre<6, 2, 8, 4>",
       "FETCH_SIZE_KB": out["FETCH_SIZE"], "WRITE_SIZE_KB": out["WRITE_SIZE"],
       "fetch_bytes_raw": out["FETCH_SIZE"]["mean_kb"] * 1024, "fetch_bytes_corrected_x2": out["FETCH_SIZE"]["mean_kb"] * 1024 * 2,
       "write_bytes": out["WRITE_SIZE"]["mean_kb"] * 1024, "algorithmic_bytes_per_launch": alg,
       "note": "history rows are read by 16-byte loads of consecutive lanes (wide coalesced stream): FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md"}
res["traffic_bytes_per_launch"] = res["fetch_bytes_corrected_x2"] + res["write_bytes"]
json.dump(res, open("gpurun_out/r01_pmc_lbfgs_pre.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pmcl_FETCH_SIZE gpurun_out/pmcl_WRITE_SIZE
