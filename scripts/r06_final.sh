#!/bin/bash
# Round 6: the ONE profile set on the closing build (reference staged in .refstage for this lease).
#   the whole GPU suite (per file, then in one process as the driver runs it), smoke, the driver's bench command, kernel
#   trace and PMC passes of the metric workload alone and of the trained-scale frame, configs[1]/[2] through the
#   unmodified scripts at scale (pixels AND gradients of the trained rows against the oracle), fuzz soak.
set -u
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export HGS_REFERENCE=$R/.refstage
rm -f gpurun_out/parity_log.jsonl gpurun_out/scale_parity.jsonl
echo "== pytest -m gpu (one process per file)"
: > gpurun_out/final_pytest_gpu.log
for f in tests/test_*gpu*.py tests/test_upstream_pins.py; do
  echo "--- $f" >> gpurun_out/final_pytest_gpu.log
  HGS_CHAIN_ITERS=1000 timeout 900 python -m pytest $f -q -m gpu -rA -s -p no:cacheprovider --durations=3 >> gpurun_out/final_pytest_gpu.log 2>&1
  echo "$f exit $?"
done
grep -aE "^(FAILED|ERROR)|passed|failed|skipped" gpurun_out/final_pytest_gpu.log | grep -v "^PASSED" | tail -30
cp gpurun_out/scale_parity.jsonl gpurun_out/final_scale_parity.jsonl 2>/dev/null
echo "== pytest -m gpu (one process, the driver's command; no reference checkout visible)"
HGS_REFERENCE=/nonexistent timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/final_pytest_gpu_one_process.log 2>&1; echo "exit $?"
tail -3 gpurun_out/final_pytest_gpu_one_process.log
echo "== smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke exit $?"
echo "== bench (the driver's command)"
S0=$(date +%s); timeout 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench exit $? in $(( $(date +%s) - S0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "stages", {k: round(v, 4) for k, v in d["stages_ms"].items()})
print("roofline frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"].get("traffic"), "cpu_baseline", (d.get("cpu_baseline") or {}).get("value"), "batched", (d.get("batched") or {}).get("value"))
for k, v in (d.get("extra") or {}).items():
    print(" extra", k, v.get("error") or (round(v.get("value", 0), 1), v.get("unit"), "ms", round(v.get("ms_per_step", 0), 3)), {kk: round(vv, 4) for kk, vv in (v.get("stages_ms") or {}).items()})
c5 = (d.get("extra") or {}).get("config5_budgeted_6gb", {})
print(" budgeted frame_ms", c5.get("frame_ms"), "occupancy", c5.get("config", {}).get("budget_occupancy_min_median_max"), "tau px", c5.get("config", {}).get("rendered_tau_px"))
PY
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/fprof -o r -- python $R/bench.py --steps 60 --warmup 30 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --no-live-pmc --schedule dropin > $R/gpurun_out/final_rocprof_dropin.log 2>&1; echo "rocprof exit $?"
timeout 100 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d /tmp/fpmc_SQ -o pmc -- $B > /dev/null 2>&1; echo "pmc SQ exit $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $c --kernel-trace -d /tmp/fpmc_$c -o pmc -- $B > /dev/null 2>&1; echo "pmc $c exit $?"
done
cd $R
HGS_SKIP_CALLS=30 python scripts/rocprof_summary.py $(ls /tmp/fprof/*.db | head -1) > gpurun_out/final_kernel_stats_dropin.txt 2>/dev/null
python scripts/pmc_summary.py SQ=$(ls /tmp/fpmc_SQ/*.db | head -1) F=$(ls /tmp/fpmc_FETCH_SIZE/*.db | head -1) W=$(ls /tmp/fpmc_WRITE_SIZE/*.db | head -1) > gpurun_out/final_pmc_summary.json 2>/dev/null; echo "pmc summary exit $?"
rm -rf /tmp/fprof /tmp/fpmc_*
head -17 gpurun_out/final_kernel_stats_dropin.txt | cut -c1-150
echo "== the trained-scale frame alone: kernel trace + SQ counters"
bash scripts/r06_pmc.sh index > gpurun_out/final_trained_pmc.txt 2>&1; head -14 gpurun_out/final_trained_pmc.txt | cut -c1-150
cp gpurun_out/r06_pmc_kernel_stats.txt gpurun_out/final_trained_kernel_stats.txt 2>/dev/null; cp gpurun_out/r06_pmc.json gpurun_out/final_trained_pmc.json 2>/dev/null
echo "== configs[1] / [2] through the unmodified scripts at scale"
HGS_PARITY_DUMP=$R/gpurun_out/parity_dump_final HGS_PARITY_DUMP_ON_FAIL=1 timeout 1500 python scripts/run_config2_config3.py > gpurun_out/final_config2_config3_scripts.log 2>&1; echo "exit $?"
grep -E "iterations/s|wall per iteration|op stages|op counters|parity|delta|gradients|GRADIENT|allocator at" gpurun_out/final_config2_config3_scripts.log | cut -c1-330
echo "== fuzz soak"
timeout 200 python tests/tools/fuzz_parity.py 40 > gpurun_out/final_fuzz_parity.json 2> /dev/null; echo "fuzz exit $?"
python -c "
import json; d=json.load(open('gpurun_out/final_fuzz_parity.json')); print({k: (len(v) if isinstance(v, list) else v) for k, v in d.items() if k in ('cases','index_mismatches','above_tolerance')})" 2>/dev/null
du -sh gpurun_out
