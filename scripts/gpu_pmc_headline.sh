cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-plan --large-batch 0"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmch_fetch -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmch_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmch_write -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmch_write.err
cd $R
python - <<'PY'
import csv, json, collections
out = {}
for name in ("fetch", "write"):
    rows = list(csv.DictReader(open(f"gpurun_out/pmch_{name}/p_counter_collection.csv")))
    v = [float(r["Counter_Value"]) for r in rows if "k_penalty" in r["Kernel_Name"]]
    out[name] = {"n": len(v), "mean_kb": sum(v) / len(v), "min_kb": min(v), "max_kb": max(v)}
fetch_b = out["fetch"]["mean_kb"] * 1024 * 2      # gfx950: FETCH_SIZE reports half of a wide coalesced stream (MI355X_MICROARCH.md, HBM)
write_b = out["write"]["mean_kb"] * 1024
res = {"command": "python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-plan --large-batch 0", "kernel": "frx::k_penalty",
       "workload": "headline (32 x 64 x kappa 16)", "FETCH_SIZE_KB": out["fetch"], "WRITE_SIZE_KB": out["write"],
       "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b, "traffic_bytes_per_launch": fetch_b + write_b,
       "algorithmic_bytes_per_launch": 1425408}
json.dump(res, open("gpurun_out/r01_pmc_headline.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf gpurun_out/pmch_fetch gpurun_out/pmch_write
