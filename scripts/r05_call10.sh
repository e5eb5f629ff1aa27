#!/bin/bash
# A/B: the tile-bin scatter with its chunk grouped by tile in LDS before the stores (HGS_TB_STAGED) against the batched
# instance-by-instance stores
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in tests/test_raster_gpu.py tests/test_product_paths_gpu.py tests/test_lod_gpu.py tests/test_properties_gpu.py; do timeout 600 python -m pytest $f -q -m gpu -rf --tb=short -p no:cacheprovider 2>&1 | tail -3; done
run() { echo "--- $1"; timeout 300 python scripts/bench_min.py 2 --no-secondary 2>&1 | tail -2; }
L=hierarchical-3d-gaussians_amd/lib/libhgs.so; cp $L /tmp/libhgs_product.so
run "staged [product]"
cp ab_variants/libhgs_unstaged.so $L; run "unstaged"
cp /tmp/libhgs_product.so $L
run "staged [product] again"
cd /tmp; timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/fprof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > /dev/null 2>&1; cd $R
python scripts/rocprof_summary.py $(ls /tmp/fprof/*.db | head -1) 2>/dev/null | head -12 | cut -c1-150
