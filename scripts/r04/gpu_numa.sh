cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
lscpu | grep -E "NUMA|Socket|Model name|Thread|Core" | head -12
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) $(cat $d/uevent 2>/dev/null | grep PCI_SLOT_NAME)"; done 2>/dev/null | head -12
python -c "
import torch; print(torch.cuda.get_device_properties(0).name, torch.cuda.device_count())"
rocm-smi --showtoponuma 2>/dev/null | head -20
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
echo "node0 cpus: $N0"; echo "node1 cpus: $N1"
echo "default:  $(timeout 100 python scripts/r03/plan_once.py 2>/dev/null)"
echo "node0:    $(taskset -c $N0 timeout 100 python scripts/r03/plan_once.py 2>/dev/null)"
[ -n "$N1" ] && echo "node1:    $(taskset -c $N1 timeout 100 python scripts/r03/plan_once.py 2>/dev/null)"
echo "node0 gaps: $(taskset -c $N0 timeout 120 python scripts/r04/round_gaps.py 32 3000 240 2>&1 | grep -E 'adj_end' | tr '\n' '|')"
[ -n "$N1" ] && echo "node1 gaps: $(taskset -c $N1 timeout 120 python scripts/r04/round_gaps.py 32 3000 240 2>&1 | grep -E 'adj_end' | tr '\n' '|')"
