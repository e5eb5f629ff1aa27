#!/bin/bash
# usage: run_timeline.sh variant... (ab_variants/libhgs_<variant>.so built from scripts/diag/k6_k7_timeline.patch)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
L=hierarchical-3d-gaussians_amd/lib/libhgs.so
for v in "$@"; do
  cp ab_variants/libhgs_$v.so $L
  for sc in ${SCENES:-metric trained}; do
    echo "=== $v $sc"
    python scripts/diag_k7_trace.py $sc gpurun_out/timeline_${v}_$sc.npz 2>&1 | grep -v amdgpu.ids
  done
done
