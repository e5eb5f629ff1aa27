#!/bin/bash
# The closing check of a round on a small GPU budget: the whole -m gpu suite, smoke, the default bench line (which now
# collects its own PMC counters) and the kernel trace + PMC summaries of the metric workload (scripts/final_prof.sh).
set -u
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl gpurun_out/scale_parity.jsonl
timeout 400 python -m pytest tests -q -m gpu -rA -s -p no:cacheprovider --timeout=300 --durations=6 > gpurun_out/final_pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -a "passed\|failed" gpurun_out/final_pytest_gpu.log | tail -2; grep -a "^FAILED\|^ERROR" gpurun_out/final_pytest_gpu.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke exit $?"
T0=$SECONDS; timeout 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench exit $? wall $((SECONDS-T0)) s"; tail -2 gpurun_out/final_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
rf = d["roofline"]
print("value", round(d["value"], 1), "traffic", rf.get("traffic"), "|", (rf.get("traffic_source") or "")[:160])
print({k: round(v["ratio"], 2) for k, v in (rf.get("traffic_by_stage") or {}).items() if v.get("ratio")})
PY
bash scripts/final_prof.sh 2>&1 | tail -8
du -sh gpurun_out
