#!/bin/bash
set -u
export TMPDIR=/tmp
run() { echo "--- $1"; timeout 300 python scripts/bench_min.py 2 --no-secondary 2>&1 | tail -2; }
L=hierarchical-3d-gaussians_amd/lib/libhgs.so; cp $L /tmp/libhgs_product.so
run "default (K6: 78 registers, 6 waves per SIMD, prefetch)"
cp ab_variants/libhgs_k6w7.so $L; run "K6 at 72 registers (7 waves), prefetch, 16 B scratch"
cp ab_variants/libhgs_k6w7np.so $L; run "K6 at 72 registers (7 waves), no prefetch, no scratch"
cp /tmp/libhgs_product.so $L
