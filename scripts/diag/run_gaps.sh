#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d /tmp/gprof -o r -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --no-live-pmc --schedule dropin > /tmp/gaps.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_gaps.py $(ls /tmp/gprof/*.db | head -1)
