# Round-6 counter passes (as round 5 - HBM traffic, VALU, dynamic FP64 counts - plus the 512-candidate batch for the knot kernels at Monte-Carlo scale, SALU and the non-FP64 share of the VALU instructions, and the leader-side instruction count of k_eval_cluster against the stage kernels; round 5 added the DYNAMIC FP64 instruction counts of k_penalty_lat: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 per wave) (run in the SAME gpurun call as the bench line they accompany; bench.py reads the result with "from_profile": true):
#   calibration       FETCH_SIZE / WRITE_SIZE of a 64 MiB coalesced copy with 8-byte and with 16-byte accesses per lane -> bytes per counted KB
#   HBM traffic       FETCH_SIZE, WRITE_SIZE (separate --pmc passes) of k_penalty_lat, k_forward_knot, k_backward_knot at the headline batch and at 1024 candidates
#   VALU utilisation  SQ counters of k_penalty_lat on both batches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -x $R/scripts/micro/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $R/scripts/micro/fetch_calib $R/scripts/micro/fetch_calib.hip
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc6_cal_$c -o p -- $R/scripts/micro/fetch_calib > /dev/null 2> $R/gpurun_out/pmc6_cal_$c.err
done
CMD="python $R/scripts/kernel_sweep.py --batches 32,512,1024 --states it60 --reps 30 --full"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc6_$c -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmc6_$c.err
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc6_valu -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmc6_valu.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES --output-format csv -d $R/gpurun_out/pmc6_fp64 -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmc6_fp64.err
cd $R
python - <<'PY'
import csv, json, collections, glob
def allrows(d):
    f = glob.glob(f"gpurun_out/{d}/**/p_counter_collection.csv", recursive=True) + glob.glob(f"gpurun_out/{d}/p_counter_collection.csv")
    return list(csv.DictReader(open(f[0])))
BYTES = 64 << 20
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for kn, tag in (("k_copy8", "8B"), ("k_copy16", "16B")):
        v = [float(r["Counter_Value"]) for r in allrows(f"pmc6_cal_{c}") if kn in r["Kernel_Name"]]
        kb = sum(v) / len(v)
        cal[f"{c}_{tag}"] = {"counter_KB_per_launch": kb, "true_bytes_per_launch": BYTES, "bytes_per_counted_byte": BYTES / (kb * 1024.0)}
res = {"command": "python scripts/kernel_sweep.py --batches 32,512,1024 --states it60 --reps 30 --full (headline batch and the same batch replicated 32x = 1024 candidates), rocprofv3 --pmc one counter set per pass",
       "calibration": cal, "calibration_note": "64 MiB coalesced copy, 8-byte and 16-byte accesses per lane (scripts/micro/fetch_calib.hip), same gpurun call; factor = true bytes / (counter x 1024)"}
f8, f16 = cal["FETCH_SIZE_8B"]["bytes_per_counted_byte"], cal["FETCH_SIZE_16B"]["bytes_per_counted_byte"]
w8, w16 = cal["WRITE_SIZE_8B"]["bytes_per_counted_byte"], cal["WRITE_SIZE_16B"]["bytes_per_counted_byte"]
kern = {}
for kname in ("k_penalty", "k_forward_knot", "k_backward_knot"):
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        g = collections.defaultdict(list)
        for r in allrows(f"pmc6_{c}"):
            if kname in r["Kernel_Name"]: g[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
        for grid, v in g.items(): per[grid][c + "_KB_raw"] = sum(v) / len(v)
    out = {}
    for grid, m in sorted(per.items()):
        fk, wk = m.get("FETCH_SIZE_KB_raw", 0.0) * 1024, m.get("WRITE_SIZE_KB_raw", 0.0) * 1024
        e = dict(m)
        if kname == "k_penalty":        # 16-byte staging loads, 8-byte stores of the partials
            e["traffic_bytes_per_launch"] = fk * f16 + wk * w8
        else:                           # mixed 8- and 16-byte loads: bracket between the two calibrations; stores: 8-byte (forward: 16-byte coefficient sweep)
            e["traffic_bytes_per_launch_range"] = [fk * min(f8, f16) + wk * min(w8, w16), fk * max(f8, f16) + wk * max(w8, w16)]
        out["grid_%d" % grid] = e
    kern[kname] = out
res["kernels"] = kern
# the one-launch evaluation (k_eval_cluster, headline batch only): loads of 8 and 16 bytes, granule stores of 8 - bracketed by the two calibrations
try:
    m = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = [float(r["Counter_Value"]) for r in allrows(f"pmc6_{c}") if "k_eval_cluster" in r["Kernel_Name"]]
        m[c + "_KB_raw"] = sum(v) / len(v)
    fk, wk = m["FETCH_SIZE_KB_raw"] * 1024, m["WRITE_SIZE_KB_raw"] * 1024
    m["traffic_bytes_per_launch_range"] = [fk * min(f8, f16) + wk * min(w8, w16), fk * max(f8, f16) + wk * max(w8, w16)]
    res["k_eval_cluster"] = m
except Exception as e:
    res["k_eval_cluster"] = {"error": repr(e)}
# the solo launch (k_eval_solo: one workgroup per candidate runs the three stage bodies): per batch size, bracketed like the knot kernels
try:
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        g = collections.defaultdict(list)
        for r in allrows(f"pmc6_{c}"):
            if "k_eval_solo" in r["Kernel_Name"]: g[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
        for grid, v in g.items(): per[grid][c + "_KB_raw"] = sum(v) / len(v)
    out = {}
    for grid, m in sorted(per.items()):
        fk, wk = m.get("FETCH_SIZE_KB_raw", 0.0) * 1024, m.get("WRITE_SIZE_KB_raw", 0.0) * 1024
        e = dict(m)
        e["traffic_bytes_per_launch_range"] = [fk * min(f8, f16) + wk * min(w8, w16), fk * max(f8, f16) + wk * max(w8, w16)]
        out["grid_%d" % grid] = e
    res["k_eval_solo"] = out
except Exception as e:
    res["k_eval_solo"] = {"error": repr(e)}
pen = kern["k_penalty"]
small = sorted(pen, key=lambda k: int(k.split("_")[1]))[0]
res["traffic_bytes_per_launch"] = pen[small]["traffic_bytes_per_launch"]      # headline launch of the penalty integrator
g = collections.defaultdict(lambda: collections.defaultdict(list))
for r in allrows("pmc6_valu"):
    if "k_penalty" in r["Kernel_Name"]: g[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
valu = {}
for grid, c in sorted(g.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    valu["grid_%d" % grid] = {"kernel_cycles": cyc, "valu_busy_frac": 4.0 * m["SQ_ACTIVE_INST_VALU"] / 1024 / cyc, "mean_waves_per_simd": 4.0 * m["SQ_WAVE_CYCLES"] / 1024 / cyc,
                              "valu_insts_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"], "salu_insts_per_wave": m.get("SQ_INSTS_SALU", 0.0) / m["SQ_WAVES"], "wait_frac_of_wave_cycles": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]}
res["valu_k_penalty_lat"] = valu
# VERDICT r5 item 2, the dynamic side: instructions per launch of the one-launch evaluation against the stage kernels that run the same bodies (headline batch).  The
# members of k_eval_cluster run the penalty integrator's samples (as k_penalty_lat does), its leaders the forward map and the adjoint (as the two knot kernels do):
# (VALU of k_eval_cluster) - (VALU of k_penalty_lat) is the leaders' count, to be held against forward + adjoint.
try:
    tot = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in allrows("pmc6_valu"):
        for kn in ("k_eval_cluster", "k_forward_knot", "k_backward_knot", "k_penalty_lat"):
            if kn in r["Kernel_Name"] and int(r["Grid_Size"]) <= 256 * 2048: tot[(kn, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    small = {}
    for (kn, grid), c in tot.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        if kn not in small or grid < small[kn][0]: small[kn] = (grid, m)
    dyn = {kn: {"grid_threads": g_, "valu_insts_per_launch": m["SQ_INSTS_VALU"], "salu_insts_per_launch": m.get("SQ_INSTS_SALU"), "waves": m["SQ_WAVES"]} for kn, (g_, m) in small.items()}
    if all(k in dyn for k in ("k_eval_cluster", "k_forward_knot", "k_backward_knot", "k_penalty_lat")):
        dyn["leaders_of_k_eval_cluster_valu"] = dyn["k_eval_cluster"]["valu_insts_per_launch"] - dyn["k_penalty_lat"]["valu_insts_per_launch"]
        dyn["forward_plus_adjoint_stage_kernels_valu"] = dyn["k_forward_knot"]["valu_insts_per_launch"] + dyn["k_backward_knot"]["valu_insts_per_launch"]
        dyn["leaders_of_k_eval_cluster_salu"] = dyn["k_eval_cluster"]["salu_insts_per_launch"] - dyn["k_penalty_lat"]["salu_insts_per_launch"]
        dyn["forward_plus_adjoint_stage_kernels_salu"] = dyn["k_forward_knot"]["salu_insts_per_launch"] + dyn["k_backward_knot"]["salu_insts_per_launch"]
    res["dynamic_instructions_headline"] = dyn
except Exception as e:
    res["dynamic_instructions_headline"] = {"error": repr(e)}
# dynamic FP64 work of the penalty integrator: instructions of a real launch at the bench state, per wave (= per sample: a lane is a sample);
# flops: FMA = 2, ADD / MUL / transcendental = 1
try:
    g = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in allrows("pmc6_fp64"):
        if "k_penalty" in r["Kernel_Name"]: g[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    fp = {}
    for grid, c in sorted(g.items()):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        w = m["SQ_WAVES"]
        fp["grid_%d" % grid] = {"add_per_wave": m["SQ_INSTS_VALU_ADD_F64"] / w, "mul_per_wave": m["SQ_INSTS_VALU_MUL_F64"] / w, "fma_per_wave": m["SQ_INSTS_VALU_FMA_F64"] / w,
                                "trans_per_wave": m["SQ_INSTS_VALU_TRANS_F64"] / w,
                                "flops_per_sample": (m["SQ_INSTS_VALU_ADD_F64"] + m["SQ_INSTS_VALU_MUL_F64"] + 2.0 * m["SQ_INSTS_VALU_FMA_F64"] + m["SQ_INSTS_VALU_TRANS_F64"]) / w}
    res["fp64_k_penalty_lat"] = fp
    for gk, fv in fp.items():                                            # the non-FP64 share of the VALU instructions, per launch class
        if gk in valu:
            f64 = fv["add_per_wave"] + fv["mul_per_wave"] + fv["fma_per_wave"] + fv["trans_per_wave"]
            valu[gk]["fp64_insts_per_wave"] = f64
            valu[gk]["non_fp64_valu_frac"] = 1.0 - f64 / valu[gk]["valu_insts_per_wave"]
except Exception as e:
    res["fp64_k_penalty_lat"] = {"error": repr(e)}
json.dump(res, open("gpurun_out/r06_pmc_headline.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
cd /tmp
cat > /tmp/dv256.py <<PY
import sys; sys.path.insert(0, "$R")
import frx_import, fast_racing_amd as frx
print(frx.dv_selftest(641, B=256, m=128, iters=160))
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc6l_$c -o p -- python /tmp/dv256.py > /dev/null 2> $R/gpurun_out/pmc6l_$c.err
done
cd $R
python - <<'PY'
import csv, json, glob
res = json.load(open("gpurun_out/r06_pmc_headline.json"))
out = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc6l_{name}/**/p_counter_collection.csv", recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if "k_lbfgs_pre" in r["Kernel_Name"]]
    v = sorted(float(r["Counter_Value"]) for r in rows)[-20:]          # the launches with a full history (bound = 128)
    out[name] = sum(v) / len(v) * 1024
f16 = res["calibration"]["FETCH_SIZE_16B"]["bytes_per_counted_byte"]; w8 = res["calibration"]["WRITE_SIZE_8B"]["bytes_per_counted_byte"]
res["k_lbfgs_pre"] = {"command": "frx_dv_selftest(n=641, B=256, m=128, iters=160)", "fetch_bytes_counted": out["FETCH_SIZE"], "write_bytes_counted": out["WRITE_SIZE"],
                      "traffic_bytes_per_launch": out["FETCH_SIZE"] * f16 + out["WRITE_SIZE"] * w8, "algorithmic_bytes_per_launch": 256 * 128 * 2 * 2 * 656 * 8,
                      "history_row_doubles": 656, "note": "history rows (n = 641: 656 doubles since round 6, 768 before) are read by 16-byte loads: FETCH_SIZE with the 16-byte calibration factor of this call"}
json.dump(res, open("gpurun_out/r06_pmc_headline.json", "w"), indent=1)
print("k_lbfgs_pre", res["k_lbfgs_pre"])
PY
rm -rf gpurun_out/pmc6_fp64 gpurun_out/pmc6l_FETCH_SIZE gpurun_out/pmc6l_WRITE_SIZE gpurun_out/pmc6_cal_FETCH_SIZE gpurun_out/pmc6_cal_WRITE_SIZE gpurun_out/pmc6_FETCH_SIZE gpurun_out/pmc6_WRITE_SIZE gpurun_out/pmc6_valu
