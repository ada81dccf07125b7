# round 6, sixth call: WHERE is host memory on the round's chain?  The round's timeline (instrumented instantiation) with the mailboxes on the device's NUMA node
# and on whatever node the caller ran (several processes: the slow mode shows in most of them), segment by segment
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 120 python scripts/r04/round_gaps.py 32 3000 240 > gpurun_out/r06_gaps_local_$i.txt 2>&1
  FRX_NUMA_ALLOC=0 timeout 120 python scripts/r04/round_gaps.py 32 3000 240 > gpurun_out/r06_gaps_anynode_$i.txt 2>&1
done
head -3 gpurun_out/r06_gaps_local_1.txt | cut -c1-200
python - <<'PY'
import re, glob
def load(f):
    rows = {}
    for l in open(f):
        m = re.match(r"\s*(.+?->.+?)\s+mean\s+([0-9.]+)\s+median\s+([0-9.]+)", l)
        if m: rows[re.sub(r"\s+", " ", m.group(1).strip())] = (float(m.group(2)), float(m.group(3)))
    hdr = open(f).readline()
    return rows, hdr
files = sorted(glob.glob("gpurun_out/r06_gaps_*.txt"))
data = {f: load(f) for f in files}
keys = list(data[files[0]][0].keys())
print("segment".ljust(62) + " ".join(f.split("r06_gaps_")[1][:-4].rjust(10) for f in files))
for k in keys:
    print(k[:60].ljust(62) + " ".join(("%.2f" % data[f][0].get(k, (float('nan'),))[0]).rjust(10) for f in files))
for f in files: print(f, data[f][1][:160].strip())
PY
