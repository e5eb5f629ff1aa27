#!/bin/bash
# Round 5: the ONE profile set on the closing build (reference staged in .refstage for this lease).
#   tests (whole GPU suite, incl. the staged-reference acceptance tests), smoke, the driver's bench command, kernel trace
#   and PMC passes of the metric workload alone, configs[1]/[2] through the unmodified scripts at scale, fuzz soak.
set -u
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export HGS_REFERENCE=$R/.refstage
rm -f gpurun_out/parity_log.jsonl gpurun_out/scale_parity.jsonl
echo "== pytest -m gpu (one process per file)"
: > gpurun_out/final_pytest_gpu.log
for f in tests/test_*gpu*.py; do
  echo "--- $f" >> gpurun_out/final_pytest_gpu.log
  HGS_CHAIN_ITERS=1000 timeout 900 python -m pytest $f -q -m gpu -rA -s -p no:cacheprovider --durations=3 >> gpurun_out/final_pytest_gpu.log 2>&1
  echo "$f exit $?"
done
grep -aE "^(FAILED|ERROR)|passed|failed|skipped" gpurun_out/final_pytest_gpu.log | grep -v "^PASSED" | tail -24
echo "== smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke exit $?"
echo "== bench (the driver's command)"
timeout 400 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "stages", {k: round(v, 4) for k, v in d["stages_ms"].items()})
print("roofline frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"].get("traffic"), "cpu_baseline", (d.get("cpu_baseline") or {}).get("value"))
for k, v in (d.get("extra") or {}).items():
    print(" extra", k, round(v.get("value", 0), 1), v.get("unit"), "ms", round(v.get("ms_per_step", 0), 3))
c5 = (d.get("extra") or {}).get("config5_budgeted_6gb", {}).get("config", {})
print(" budgeted occupancy", c5.get("budget_occupancy_min_median_max"), "finer", c5.get("rows_needed_one_step_finer_over_budget"), "tau px", c5.get("rendered_tau_px"))
PY
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin"
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/fprof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > $R/gpurun_out/final_rocprof_dropin.log 2>&1; echo "rocprof exit $?"
timeout 100 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d /tmp/fpmc_SQ -o pmc -- $B > /dev/null 2>&1; echo "pmc SQ exit $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $c --kernel-trace -d /tmp/fpmc_$c -o pmc -- $B > /dev/null 2>&1; echo "pmc $c exit $?"
done
cd $R
python scripts/rocprof_summary.py $(ls /tmp/fprof/*.db | head -1) > gpurun_out/final_kernel_stats_dropin.txt 2>/dev/null
python scripts/pmc_summary.py SQ=$(ls /tmp/fpmc_SQ/*.db | head -1) F=$(ls /tmp/fpmc_FETCH_SIZE/*.db | head -1) W=$(ls /tmp/fpmc_WRITE_SIZE/*.db | head -1) > gpurun_out/final_pmc_summary.json 2>/dev/null; echo "pmc summary exit $?"
rm -rf /tmp/fprof /tmp/fpmc_*
head -16 gpurun_out/final_kernel_stats_dropin.txt | cut -c1-150
echo "== configs[1] / [2] through the unmodified scripts at scale"
timeout 1200 python scripts/run_config2_config3.py > gpurun_out/final_config2_config3_scripts.log 2>&1; echo "exit $?"
grep -E "iterations/s|wall per iteration|op stages|op counters|parity|delta" gpurun_out/final_config2_config3_scripts.log | cut -c1-330
echo "== fuzz soak"
timeout 150 python tests/tools/fuzz_parity.py 40 > gpurun_out/final_fuzz_parity.json 2> /dev/null; echo "fuzz exit $?"
python -c "
import json; d=json.load(open('gpurun_out/final_fuzz_parity.json')); print({k: (len(v) if isinstance(v, list) else v) for k, v in d.items() if k in ('cases','index_mismatches','above_tolerance')})" 2>/dev/null
du -sh gpurun_out
