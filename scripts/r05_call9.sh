#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in tests/test_raster_gpu.py tests/test_product_paths_gpu.py tests/test_lod_gpu.py tests/test_scale_parity_gpu.py; do timeout 600 python -m pytest $f -q -m gpu -rf --tb=short -p no:cacheprovider 2>&1 | tail -3; done
run() { echo "--- $1"; timeout 300 python scripts/bench_min.py 2 --no-secondary 2>&1 | tail -2; }
L=hierarchical-3d-gaussians_amd/lib/libhgs.so; cp $L /tmp/libhgs_product.so
run "layout 1: two groups of 128 whole rows (24 KB)  [product]"
cd /tmp; timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/f1 -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > /dev/null 2>&1; cd $R
python scripts/pmc_summary.py F=$(ls /tmp/f1/*.db | head -1) 2>/dev/null | grep -A2 "preprocess_fwd" | head -4
cp ab_variants/libhgs_lay0.so $L; run "layout 0: half rows (24 KB)"
cp ab_variants/libhgs_lay2.so $L; run "layout 2: 256 whole rows (48 KB)"
cp /tmp/libhgs_product.so $L
