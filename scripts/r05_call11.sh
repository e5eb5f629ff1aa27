#!/bin/bash
# Experiment: does the count kernel's time depend on the idle workgroups of its 8x over-provisioned grid?
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 1 4; do
  echo "--- HGS_TB_GRID_DIV=$d"
  cd /tmp; HGS_TB_GRID_DIV=$d timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/fprof$d -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > /dev/null 2>&1; cd $R
  python scripts/rocprof_summary.py $(ls /tmp/fprof$d/*.db | head -1) 2>/dev/null | grep -E "tb_|duplicate" | cut -c1-150
done
