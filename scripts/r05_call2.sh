#!/bin/bash
# Round 5, second lease: K1 anatomy, the GPU suite file by file (a crash in one file must not hide the others), bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== K1 anatomy (us per launch, stage1 only)"
timeout 900 python scripts/diag_k1_anatomy.py cs0 x1 x2 x3 x4 x8 x16 x27 x31 2>&1 | tee gpurun_out/k1_anatomy.txt
echo "== pytest -m gpu, one process per file"
rm -f gpurun_out/parity_log.jsonl gpurun_out/scale_parity.jsonl gpurun_out/pytest_gpu.log
for f in tests/test_*gpu*.py; do
  echo "--- $f" >> gpurun_out/pytest_gpu.log
  timeout 900 python -m pytest $f -q -m gpu -rf --tb=short -p no:cacheprovider >> gpurun_out/pytest_gpu.log 2>&1
  echo "$f exit $?"
done
grep -E "^(FAILED|ERROR)|passed|failed|^--- " gpurun_out/pytest_gpu.log | tail -60
echo "== bench: product"
timeout 600 python scripts/bench_min.py 2 2>&1 | tail -3
