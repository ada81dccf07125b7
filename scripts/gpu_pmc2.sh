cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc2_$name -o p -- $CMD > /dev/null 2> $R/gpurun_out/pmc2_$name.err; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run b SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_LDS
cd $R
python - <<'PY'
import csv, collections
for name in ("a","b"):
    rows=list(csv.DictReader(open(f"gpurun_out/pmc2_{name}/p_counter_collection.csv")))
    agg=collections.defaultdict(list)
    for r in rows:
        if "k_lbfgs_pre" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        v=sorted(v)[-200:]   # the long (bound=128) advances
        print(name,k,len(v),"mean %.5g"%(sum(v)/len(v)))
PY
