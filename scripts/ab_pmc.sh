#!/bin/bash
# Tuning aid (GPU box): SQ counters of the two compositing kernels for each variant built by scripts/ab_build.sh.
L=hierarchical-3d-gaussians_amd/lib/libhgs.so
R=$GRAFT_REPO_ROOT
cp $L /tmp/libhgs_product.so
export TMPDIR=/tmp
mkdir -p gpurun_out
for name in "$@"; do
  echo "=== variant $name"
  [ "$name" != product ] && cp ab_variants/libhgs_$name.so $L
  rm -rf /tmp/abpmc_$name
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d /tmp/abpmc_$name -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > /tmp/abpmc_$name.log 2>&1)
  python scripts/pmc_summary.py SQ=$(ls /tmp/abpmc_$name/*.db | head -1) > gpurun_out/abpmc_$name.json
  python - <<PY
import json
d = json.load(open("gpurun_out/abpmc_$name.json"))
for k, c in d.items():
    if "render" in k:
        w = c["SQ_WAVES"]
        print(k[-40:], {x: round(c[x] / w, 1) for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY") if x in c})
PY
  cp /tmp/libhgs_product.so $L
done
