# VALU utilisation of k_penalty (headline launches and the replicated 1024-candidate batch of the same bench command).
# Counters only (--kernel-trace + --pmc), separate from any other profiling pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-plan --large-batch 1024"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcv -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmcv.err
cd $R
python - <<'PY'
import csv, json, collections
rows = [r for r in csv.DictReader(open("gpurun_out/pmcv/p_counter_collection.csv")) if "k_penalty" in r["Kernel_Name"]]
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: by[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"command": "python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-plan --large-batch 1024", "kernel": "frx::k_penalty", "simds": 1024, "launch_classes": {}}
for grid, c in sorted(by.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    m["launches"] = len(next(iter(c.values())))
    # rocprof's VALUBusy: 4 cycles per wave64 VALU instruction slot on a SIMD16; SQ_WAVE_CYCLES counts in units of 4 cycles
    cyc = m["GRBM_GUI_ACTIVE"] / 8                     # counters are summed over the 8 XCDs
    m["kernel_cycles"] = cyc
    m["valu_busy_frac"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / 1024 / cyc
    m["mean_waves_per_simd"] = 4.0 * m["SQ_WAVE_CYCLES"] / 1024 / cyc
    m["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
    m["wait_frac_of_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    res["launch_classes"]["grid_%d" % grid] = m
json.dump(res, open("gpurun_out/r01_pmc_valu.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pmcv
