# round 3, call 1: parity suite with the resident kernel's own verdicts (no retry), direction pin, 8-shard config[3]; baseline round budget
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=8 > gpurun_out/tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/tests.log | tail -3; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/tests.log | head -30
tail -15 gpurun_out/tests.log
cat gpurun_out/direction_pin_s*.json 2>/dev/null | tr -d '\n '; echo
for f in gpurun_out/share_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print({k:d[k] for k in d if k not in ('stalled_with_other_label',)})
PY
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python scripts/resident_profile.py 32 64 16 3000 > gpurun_out/r03_base_budget_B32.json 2>&1; head -60 gpurun_out/r03_base_budget_B32.json | tr -d '\n '; echo
