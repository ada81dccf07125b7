cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/scripts/kernel_sweep.py --batches 4096 --states it60 --reps 5"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmc_$name.err; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
python - <<'PY'
import csv, glob, collections
for name in ("sq1","sq2","fetch","write"):
    files = glob.glob(f"gpurun_out/pmc_{name}/**/*counter_collection.csv", recursive=True)
    if not files: print(name, "no csv"); continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if "k_penalty" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        v = v[len(v)//2:]   # the B=4096 launches come last
        print(name, k, "n=%d" % len(v), "mean=%.4g" % (sum(v)/len(v)))
PY
