#!/bin/bash
# Round 5, fifth lease (reference staged): configs[1]/[2] at scale on the fixed K8 / speculation, the changed tests,
# K6 / K7 against the number of tiles.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export HGS_REFERENCE=$PWD/.refstage
echo "== configs[1] / [2] at scale"
timeout 1500 python scripts/run_config2_config3.py > gpurun_out/r05_config2_config3_scripts.log 2>&1; echo "exit $?"
grep -E "^==|^---|iterations/s|wall per iteration|op stages|op counters|allocator|PSNR|parity|FAILED|Error" gpurun_out/r05_config2_config3_scripts.log | cut -c1-400
echo "== pytest"
rm -f gpurun_out/parity_log.jsonl gpurun_out/scale_parity.jsonl gpurun_out/pytest_gpu.log
for f in "tests/test_reference_on_gpu.py -k glue" tests/test_raster_gpu.py tests/test_product_paths_gpu.py tests/test_lod_gpu.py tests/test_bench_contract_gpu.py; do
  echo "--- $f" >> gpurun_out/pytest_gpu.log
  timeout 900 python -m pytest $f -q -m gpu -rf --tb=short -p no:cacheprovider >> gpurun_out/pytest_gpu.log 2>&1
  echo "$f exit $?"
done
grep -E "^(FAILED|ERROR)|passed|failed|^--- |Error" gpurun_out/pytest_gpu.log | tail -30
echo "== bench (default)"
timeout 600 python scripts/bench_min.py 2 2>&1 | tail -3
for lim in 4096 2048 1024; do
  echo "== bench, K6 / K7 limited to the first $lim tiles"
  HGS_RENDER_GRID_LIMIT=$lim timeout 600 python scripts/bench_min.py 1 --no-secondary 2>&1 | tail -2
done
