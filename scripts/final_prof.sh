#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the metric workload alone (bench.py --no-extras: nothing else launches the kernels)
set -u
R=$GRAFT_REPO_ROOT; cd /tmp; mkdir -p $R/gpurun_out; export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin"
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/fprof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > $R/gpurun_out/final_rocprof_dropin.log 2>&1; echo "rocprof exit $?"
timeout 100 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d /tmp/fpmc_SQ -o pmc -- $B > /dev/null 2>&1; echo "pmc SQ exit $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $c --kernel-trace -d /tmp/fpmc_$c -o pmc -- $B > /dev/null 2>&1; echo "pmc $c exit $?"
done
cd $R
python scripts/rocprof_summary.py $(ls /tmp/fprof/*.db | head -1) > gpurun_out/final_kernel_stats_dropin.txt 2>/dev/null
python scripts/pmc_summary.py SQ=$(ls /tmp/fpmc_SQ/*.db | head -1) F=$(ls /tmp/fpmc_FETCH_SIZE/*.db | head -1) W=$(ls /tmp/fpmc_WRITE_SIZE/*.db | head -1) > gpurun_out/final_pmc_summary.json 2>/dev/null; echo "pmc summary exit $?"
rm -rf /tmp/fprof /tmp/fpmc_*
head -14 gpurun_out/final_kernel_stats_dropin.txt | cut -c1-140
