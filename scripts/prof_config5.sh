#!/bin/bash
# Tuning aid (GPU box): kernel times of the configs[4] extra (50 M-node hierarchy: cut + weights + 4K render per frame).
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/prof_c5
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c5 -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-stage-timing --extras config5_50m_4k_render > $R/gpurun_out/rocprof_c5.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py $(ls gpurun_out/prof_c5/*.db | head -1) > gpurun_out/kernel_stats_config5.txt 2>/dev/null; head -40 gpurun_out/kernel_stats_config5.txt
tail -c 1500 gpurun_out/rocprof_c5.log
rm -rf gpurun_out/prof_c5
