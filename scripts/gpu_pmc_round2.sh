# Round-2 counter passes of the penalty integrator (run in the SAME gpurun call as the bench line they accompany):
#   HBM traffic per launch  FETCH_SIZE, WRITE_SIZE in separate --pmc runs; FETCH_SIZE doubled (gfx950 counts a 16-byte-per-lane coalesced
#                           stream at half its bytes, MI355X_MICROARCH.md "HBM"; the kernel stages with 16-byte loads since round 2)
#   VALU utilisation        SQ counters on the replicated 1024-candidate batch and on the headline batch, for both forms of the kernel
#                           (latency form = default, throughput form = FRX_PENALTY_FORM=thr)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/scripts/kernel_sweep.py --batches 32,1024 --states it60 --reps 30 --full"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc2_$c -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmc2_$c.err
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2_valu -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmc2_valu.err
FRX_PENALTY_FORM=thr timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2_valu4 -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmc2_valu4.err
cd $R
python - <<'PY'
import csv, json, collections
def rows(d): return [r for r in csv.DictReader(open(f"gpurun_out/{d}/p_counter_collection.csv")) if "k_penalty" in r["Kernel_Name"]]
res = {"command": "python scripts/kernel_sweep.py --batches 32,1024 --states it60 --reps 30 --full (headline batch and the same batch replicated 32x = 1024 candidates)", "kernel": "frx::k_penalty"}
by = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    g = collections.defaultdict(list)
    for r in rows(f"pmc2_{c}"): g[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    for grid, v in g.items(): by[grid][c + "_KB"] = sum(v) / len(v)
traffic = {}
for grid, m in sorted(by.items()):
    fb, wb = m["FETCH_SIZE_KB"] * 1024 * 2, m["WRITE_SIZE_KB"] * 1024
    traffic["grid_%d" % grid] = {"FETCH_SIZE_KB_raw": m["FETCH_SIZE_KB"], "WRITE_SIZE_KB": m["WRITE_SIZE_KB"], "fetch_bytes_x2": fb, "write_bytes": wb, "traffic_bytes_per_launch": fb + wb}
res["traffic"] = traffic
# the two knot kernels of the same runs (the sweep's --full leg): RAW counters, no correction - these kernels mix 8- and 16-byte loads, which the
# guide's x2 rule for FETCH_SIZE is not calibrated for, so the figures are reported as they come and bench.py does not turn them into roofline.traffic
def krows(d, name): return [r for r in csv.DictReader(open(f"gpurun_out/{d}/p_counter_collection.csv")) if name in r["Kernel_Name"]]
raw = {}
for kname in ("k_forward_knot", "k_backward_knot"):
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        g = collections.defaultdict(list)
        for r in krows(f"pmc2_{c}", kname): g[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
        for grid, v in g.items(): per["grid_%d" % grid][c + "_KB_raw"] = sum(v) / len(v)
    raw[kname] = dict(per)
res["knot_kernels_raw_uncalibrated"] = raw
smallest = min(by)
res["traffic_bytes_per_launch"] = traffic["grid_%d" % smallest]["traffic_bytes_per_launch"]          # headline launch
for name in ("pmc2_valu", "pmc2_valu4"):
    g = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(name): g[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for grid, c in sorted(g.items()):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        out["grid_%d" % grid] = {"kernel_cycles": cyc, "valu_busy_frac": 4.0 * m["SQ_ACTIVE_INST_VALU"] / 1024 / cyc, "mean_waves_per_simd": 4.0 * m["SQ_WAVE_CYCLES"] / 1024 / cyc,
                                 "valu_insts_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"], "wait_frac_of_wave_cycles": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]}
    res["valu_latency_form_148_vgprs_default" if name == "pmc2_valu" else "valu_throughput_form_126_vgprs"] = out
json.dump(res, open("gpurun_out/r02_pmc_headline.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pmc2_FETCH_SIZE gpurun_out/pmc2_WRITE_SIZE gpurun_out/pmc2_valu gpurun_out/pmc2_valu4
