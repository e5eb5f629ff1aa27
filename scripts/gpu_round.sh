#!/bin/bash
# One gpurun call.  Sections are picked with SECTIONS="tests smoke micro bench prof pmc next" (default: all but micro).
# Outputs land under gpurun_out/ (merged back by gpurun); copy what is to be judged into profiles/.
set -u
S=" ${SECTIONS:-tests smoke bench prof pmc next} "
has() { [[ "$S" == *" $1 "* ]]; }
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if has tests; then
  rm -f gpurun_out/parity_log.jsonl gpurun_out/scale_parity.jsonl
  echo "== pytest -m gpu ${PYTEST_ARGS:-}"
  timeout 1500 python -m pytest tests -q -m gpu -rA -p no:cacheprovider --durations=8 ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?"; tail -25 gpurun_out/pytest_gpu.log
fi
if has smoke; then
  echo "== smoke"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
fi
if has micro; then
  echo "== VALU issue-rate microbenchmark"
  timeout 300 scripts/microbench/valu_issue_bench > gpurun_out/microbench_valu_issue.txt 2>&1; echo "micro exit $?"; cat gpurun_out/microbench_valu_issue.txt
fi
if has bench; then
  echo "== bench (driver's command line)"
  timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
fi
if has prof; then
  echo "== rocprofv3 kernel trace: drop-in schedule, then batched schedule"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dropin -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > $R/gpurun_out/rocprof_dropin.log 2>&1; echo "rocprof dropin exit $?"
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_batched -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule batched > $R/gpurun_out/rocprof_batched.log 2>&1; echo "rocprof batched exit $?"
  cd $R
  python scripts/rocprof_summary.py $(ls gpurun_out/prof_dropin/*.db | head -1) > gpurun_out/kernel_stats_dropin.txt 2>/dev/null; head -22 gpurun_out/kernel_stats_dropin.txt
  python scripts/rocprof_summary.py $(ls gpurun_out/prof_batched/*.db | head -1) > gpurun_out/kernel_stats_batched.txt 2>/dev/null
  rm -rf gpurun_out/prof_dropin gpurun_out/prof_batched      # the summaries are what is kept (gpurun copies back <= 64 MiB)
fi
if has pmc; then
  echo "== PMC passes (drop-in schedule)"
  cd /tmp
  B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin"
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $R/gpurun_out/pmc_SQ -o pmc -- $B > $R/gpurun_out/pmc_SQ.log 2>&1; echo "pmc SQ exit $?"
  timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_SQ2 -o pmc -- $B > $R/gpurun_out/pmc_SQ2.log 2>&1; echo "pmc SQ2 exit $?"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$c -o pmc -- $B > $R/gpurun_out/pmc_$c.log 2>&1; echo "pmc $c exit $?"
  done
  cd $R
  python scripts/pmc_summary.py SQ=gpurun_out/pmc_SQ/pmc_results.db SQ2=gpurun_out/pmc_SQ2/pmc_results.db F=gpurun_out/pmc_FETCH_SIZE/pmc_results.db W=gpurun_out/pmc_WRITE_SIZE/pmc_results.db > gpurun_out/pmc_summary.json 2>/dev/null; echo "pmc summary exit $?"
  rm -rf gpurun_out/pmc_SQ gpurun_out/pmc_SQ2 gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
fi
du -sh gpurun_out 2>/dev/null
if has next; then
  echo "== bench_next (SURVEY 8f rows)"
  timeout 600 python scripts/bench_next.py > gpurun_out/bench_next.jsonl 2> gpurun_out/bench_next.err; echo "bench_next exit $?"; cat gpurun_out/bench_next.jsonl; tail -3 gpurun_out/bench_next.err
fi
if has dp2; then
  echo "== 2 ranks sharing the one GPU (gloo): bench + DP tests"
  HGS_DP_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --no-stage-timing --no-secondary > gpurun_out/bench_dp2_gloo.json 2> gpurun_out/bench_dp2.err; echo "dp2 exit $?"; cat gpurun_out/bench_dp2_gloo.json | head -c 600; tail -3 gpurun_out/bench_dp2.err
fi
if has dp2; then
  echo "== same, gradient exchange by the direct peer-pointer all-reduce (hgs_p2p_*)"
  HGS_DP_BACKEND=gloo HGS_DP_ALLREDUCE=direct timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --no-stage-timing --no-secondary > gpurun_out/bench_dp2_direct.json 2> gpurun_out/bench_dp2_direct.err; echo "dp2 direct exit $?"; cat gpurun_out/bench_dp2_direct.json | head -c 400; tail -2 gpurun_out/bench_dp2_direct.err
fi
nproc; free -g | head -2
