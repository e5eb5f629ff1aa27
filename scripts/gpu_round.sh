#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel stats.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl
echo "== pytest -m gpu" 
timeout 1200 python -m pytest tests -q -m gpu -rA -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -40 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -5 gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
for v in ${VARIANTS:-3 2}; do
  timeout 300 python bench.py --steps 20 --warmup 5 --variant $v --no-cpu-baseline > gpurun_out/bench_v$v.json 2>> gpurun_out/bench.err; cat gpurun_out/bench_v$v.json
done
echo "== rocprof"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stage-timing > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof exit $?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timing > $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ.log 2>&1; echo "pmc SQ exit $?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ2 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timing > $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ2.log 2>&1; echo "pmc SQ2 exit $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timing > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1; echo "pmc $c exit $?"
done
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof gpurun_out/pmc_* | head -30; nproc; free -g | head -2
echo "== bench k=1"
timeout 300 python bench.py --steps 30 --warmup 5 --views-per-step 1 --no-cpu-baseline > gpurun_out/bench_k1.json 2>> gpurun_out/bench.err; cat gpurun_out/bench_k1.json | head -c 400; echo
echo "== bench_next (SURVEY 8f rows)"
timeout 600 python scripts/bench_next.py > gpurun_out/bench_next.jsonl 2> gpurun_out/bench_next.err; echo "bench_next exit $?"; cat gpurun_out/bench_next.jsonl; tail -3 gpurun_out/bench_next.err
echo "== rocprof, single-stream schedule (per-kernel times without co-running kernels)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_serial -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stage-timing --no-stream-overlap > /dev/null 2>&1; echo "rocprof serial exit $?"
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py $(ls gpurun_out/prof_serial/*.db | head -1) > gpurun_out/kernel_stats_serial.txt 2>/dev/null
echo "== PMC summary"
python scripts/pmc_summary.py SQ=gpurun_out/pmc_SQ/pmc_results.db SQ2=gpurun_out/pmc_SQ2/pmc_results.db F=gpurun_out/pmc_FETCH_SIZE/pmc_results.db W=gpurun_out/pmc_WRITE_SIZE/pmc_results.db > gpurun_out/pmc_summary.json 2>/dev/null; echo "pmc summary exit $?"
python scripts/rocprof_summary.py $(ls gpurun_out/prof/*.db | head -1) > gpurun_out/kernel_stats.txt 2>/dev/null; head -25 gpurun_out/kernel_stats.txt
