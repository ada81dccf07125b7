cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in "$@"; do
  FRX_DV_GEOM=$g timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/geom -o g -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/geom.json 2> $R/gpurun_out/geom.err
  echo "GEOM $g: $(grep lbfgs_pre $R/gpurun_out/geom/g_kernel_stats.csv | cut -d, -f2-4 | head -1)  $(python -c "
import json; d=json.loads(open('$R/gpurun_out/geom.json').read()); print('plan_ms %.1f rounds %d us/round %.1f' % (d['plan_ms'], d['plan_rounds'], 1e3*d['plan_ms']/d['plan_rounds']))")"
  rm -rf $R/gpurun_out/geom
done
