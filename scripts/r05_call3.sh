#!/bin/bash
# Round 5, third lease: split K1 -- parity tests, then split vs fused on the same box, kernel trace.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest (raster, product paths, lod, scale parity, train)"
rm -f gpurun_out/parity_log.jsonl gpurun_out/scale_parity.jsonl gpurun_out/pytest_gpu.log
for f in tests/test_raster_gpu.py tests/test_product_paths_gpu.py tests/test_lod_gpu.py tests/test_scale_parity_gpu.py tests/test_train_gpu.py tests/test_raw_gpu.py tests/test_residency_gpu.py tests/test_dp_step_gpu.py; do
  echo "--- $f" >> gpurun_out/pytest_gpu.log
  timeout 900 python -m pytest $f -q -m gpu -rf --tb=short -p no:cacheprovider >> gpurun_out/pytest_gpu.log 2>&1
  echo "$f exit $?"
done
grep -E "^(FAILED|ERROR)|passed|failed|^--- |Error" gpurun_out/pytest_gpu.log | tail -40
echo "== bench: split K1 (default)"
timeout 600 python scripts/bench_min.py 2 2>&1 | tail -3
echo "== bench: fused K1 (HGS_K1_FUSED=1)"
HGS_K1_FUSED=1 timeout 600 python scripts/bench_min.py 2 2>&1 | tail -3
echo "== rocprofv3 kernel trace (split, drop-in)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dropin -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > $R/gpurun_out/rocprof_dropin.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py $(ls gpurun_out/prof_dropin/*.db | head -1) > gpurun_out/kernel_stats_dropin.txt 2>/dev/null; head -16 gpurun_out/kernel_stats_dropin.txt
rm -rf gpurun_out/prof_dropin
