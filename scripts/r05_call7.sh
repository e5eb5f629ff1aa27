#!/bin/bash
# occupancy vs rounds: K6, K7 and the per-tile depth sort with the workgroups per CU capped by dynamic LDS, and the
# 64-register builds (8 waves per SIMD)
set -u
export TMPDIR=/tmp
run() { echo "--- $1"; shift; env "$@" timeout 300 python scripts/bench_min.py 1 --no-secondary 2>&1 | tail -1 | python -c "
import sys,re,ast
l=sys.stdin.read(); d=ast.literal_eval(l[l.index('{'):l.rindex('}')+1]); print({k: d[k] for k in ('tile_depth_sort','render_fwd','render_bwd')}, 'sum', l.split('sum')[-1].strip())"; }
run "default" A=1
run "K6 5 waves/SIMD (8 KB)" HGS_K6_DYN_LDS=4800
run "K6 4 waves/SIMD (10 KB)" HGS_K6_DYN_LDS=6800
run "K6 3 waves/SIMD (13 KB)" HGS_K6_DYN_LDS=10200
run "K7 3 waves/SIMD" HGS_K7_DYN_LDS=5000
run "sort 6 waves/SIMD" HGS_SORT_DYN_LDS=20000
run "sort 5 waves/SIMD" HGS_SORT_DYN_LDS=26000
run "sort 4 waves/SIMD" HGS_SORT_DYN_LDS=34000
L=hierarchical-3d-gaussians_amd/lib/libhgs.so; cp $L /tmp/libhgs_product.so
cp ab_variants/libhgs_k6w8.so $L; run "K6 64 registers (8 waves/SIMD, 44 B scratch)" A=1
cp ab_variants/libhgs_sw8.so $L; run "sort 64 registers (8 waves/SIMD, 40 B scratch)" A=1
cp /tmp/libhgs_product.so $L
