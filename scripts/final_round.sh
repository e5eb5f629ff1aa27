#!/bin/bash
# The round's ONE profile set on the final build, sized for a small GPU budget (every step under its own timeout;
# ~6 minutes in all).  Outputs: gpurun_out/final_* (summaries only -- the rocprof databases are deleted).
set -u
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl gpurun_out/scale_parity.jsonl
echo "== pytest -m gpu"
timeout 400 python -m pytest tests -q -m gpu -rA -s -p no:cacheprovider --timeout=300 --durations=6 > gpurun_out/final_pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -a "passed\|failed" gpurun_out/final_pytest_gpu.log | tail -2
echo "== smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke exit $?"
echo "== bench"
timeout 240 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench exit $?"
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
timeout 150 bash scripts/prof_config5_budgeted.sh > /dev/null 2>&1; mv gpurun_out/kernel_stats_config5_budgeted.txt gpurun_out/final_kernel_stats_config5_budgeted.txt 2>/dev/null
timeout 120 python tests/tools/fuzz_parity.py 30 > gpurun_out/final_fuzz_parity.json 2> /dev/null; echo "fuzz exit $?"
du -sh gpurun_out
