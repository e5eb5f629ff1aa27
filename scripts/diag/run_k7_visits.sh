#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
L=hierarchical-3d-gaussians_amd/lib/libhgs.so
for m in 1 2; do
  cp ab_variants/libhgs_tv$m.so $L
  per=1; [ $m = 2 ] && per=5
  for lim in 1024 4096 0; do
    HGS_GRID_LIMIT=$lim python scripts/diag_k7_visits.py $per 2>&1 | grep -v amdgpu.ids | head -8
  done
done
