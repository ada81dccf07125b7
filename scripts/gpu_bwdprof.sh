cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
env $v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb -o pb -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch 0 > $R/gpurun_out/pb.json 2> /dev/null
echo "$v: $(python - <<PY
import csv, json
d=json.load(open("$R/gpurun_out/pb.json"))
out=["us/round %.1f" % (1e3*d["plan_ms"]/d["plan_rounds"])]
for r in csv.DictReader(open("$R/gpurun_out/pb/pb_kernel_stats.csv")):
    if "frx" in r["Name"]: out.append(r["Name"].split("(")[0].replace("frx::","").replace("void ","")[:16]+" %.1f" % (float(r["AverageNs"])/1e3))
print(" | ".join(out))
PY
)"
rm -rf $R/gpurun_out/pb
done
