#!/bin/bash
# Round 5, fourth lease (reference staged in .refstage): configs[1]/[2] through the unmodified scripts at scale, then the
# tests that failed or changed.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export HGS_REFERENCE=$PWD/.refstage
nproc; free -g | head -2
echo "== configs[1] / [2] at scale"
timeout 1500 python scripts/run_config2_config3.py > gpurun_out/r05_config2_config3_scripts.log 2>&1; echo "exit $?"
tail -60 gpurun_out/r05_config2_config3_scripts.log
echo "== pytest: scale parity, product paths, glue on the HIP op"
rm -f gpurun_out/parity_log.jsonl gpurun_out/scale_parity.jsonl gpurun_out/pytest_gpu.log
for f in tests/test_scale_parity_gpu.py tests/test_product_paths_gpu.py "tests/test_reference_on_gpu.py -k glue"; do
  echo "--- $f" >> gpurun_out/pytest_gpu.log
  timeout 900 python -m pytest $f -q -m gpu -rf --tb=short -p no:cacheprovider >> gpurun_out/pytest_gpu.log 2>&1
  echo "$f exit $?"
done
grep -E "^(FAILED|ERROR)|passed|failed|^--- |Error" gpurun_out/pytest_gpu.log | tail -30
