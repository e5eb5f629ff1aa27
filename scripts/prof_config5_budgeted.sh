#!/bin/bash
# Tuning aid (GPU box): kernel times of the VRAM-budgeted configs[4] loop (bench.py extra config5_budgeted_6gb).
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_c5b
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5b -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-stage-timing --extras config5_budgeted_6gb > $R/gpurun_out/rocprof_c5b.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py $(ls /tmp/prof_c5b/*.db | head -1) > gpurun_out/kernel_stats_config5_budgeted.txt 2>/dev/null; head -30 gpurun_out/kernel_stats_config5_budgeted.txt
rm -rf /tmp/prof_c5b
