#!/bin/bash
# Round 5, first lease: smoke + full GPU suite on the new kernels, then new build vs the round-4 library on the same box.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=hierarchical-3d-gaussians_amd/lib/libhgs.so
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest -m gpu"
rm -f gpurun_out/parity_log.jsonl gpurun_out/scale_parity.jsonl
timeout 1500 python -m pytest tests -q -m gpu -rA -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -30
echo "== bench: new build"
timeout 600 python scripts/bench_min.py 2 2>&1 | tail -3
echo "== bench: round-4 library"
cp $L /tmp/libhgs_new.so; cp ab_variants/libhgs_r04.so $L
timeout 600 python scripts/bench_min.py 2 2>&1 | tail -3
cp /tmp/libhgs_new.so $L
echo "== rocprofv3 kernel trace (new build, drop-in)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dropin -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > $R/gpurun_out/rocprof_dropin.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py $(ls gpurun_out/prof_dropin/*.db | head -1) > gpurun_out/kernel_stats_dropin.txt 2>/dev/null; head -22 gpurun_out/kernel_stats_dropin.txt
rm -rf gpurun_out/prof_dropin
