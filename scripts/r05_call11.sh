#!/bin/bash
# K3 with the loads of its superblock prefixes issued together: tests, then stage times
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in tests/test_raster_gpu.py tests/test_product_paths_gpu.py tests/test_lod_gpu.py; do timeout 600 python -m pytest $f -q -m gpu -rf --tb=short -p no:cacheprovider 2>&1 | tail -2; done
timeout 300 python scripts/bench_min.py 3 --no-secondary 2>&1 | tail -2
cd /tmp; timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/fprof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > /dev/null 2>&1; cd $R
python scripts/rocprof_summary.py $(ls /tmp/fprof/*.db | head -1) 2>/dev/null | grep -E "tb_|duplicate|depth_sort" | cut -c1-150
