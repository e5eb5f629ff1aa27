#!/bin/bash
# The end of round 5's closing set, after the one-line guard on K3's sharing (grids above 4 096 workgroups are left alone):
# the suites that touch the binning again, then the driver's bench command, the kernel trace and the PMC passes on THAT
# build (the whole-suite log, the at-scale run through the unmodified scripts and the fuzz soak are those of
# scripts/r05_final.sh one commit earlier).
set -u
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/tail_pytest_gpu.log
for f in tests/test_raster_gpu.py tests/test_product_paths_gpu.py tests/test_lod_gpu.py tests/test_residency_gpu.py; do
  echo "--- $f" >> gpurun_out/tail_pytest_gpu.log
  timeout 200 python -m pytest $f -q -m gpu -rA -p no:cacheprovider >> gpurun_out/tail_pytest_gpu.log 2>&1; echo "$f exit $?"
done
grep -aE "passed|failed" gpurun_out/tail_pytest_gpu.log | grep -v "^PASSED"
timeout 200 python bench.py > gpurun_out/tail_bench.json 2> gpurun_out/tail_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/tail_bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "stages", {k: round(v, 4) for k, v in d["stages_ms"].items()})
print("roofline frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"].get("traffic"), "sha", d["kernel_source_sha"])
for k, v in (d.get("extra") or {}).items():
    print(" extra", k, round(v.get("value", 0), 1), v.get("unit"), "ms", round(v.get("ms_per_step", 0), 3))
PY
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin"
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats -d /tmp/fprof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --schedule dropin > /dev/null 2>&1; echo "rocprof exit $?"
timeout 60 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d /tmp/fpmc_SQ -o pmc -- $B > /dev/null 2>&1; echo "pmc SQ exit $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 60 rocprofv3 --pmc $c --kernel-trace -d /tmp/fpmc_$c -o pmc -- $B > /dev/null 2>&1; echo "pmc $c exit $?"
done
cd $R
python scripts/rocprof_summary.py $(ls /tmp/fprof/*.db | head -1) > gpurun_out/tail_kernel_stats_dropin.txt 2>/dev/null
python scripts/pmc_summary.py SQ=$(ls /tmp/fpmc_SQ/*.db | head -1) F=$(ls /tmp/fpmc_FETCH_SIZE/*.db | head -1) W=$(ls /tmp/fpmc_WRITE_SIZE/*.db | head -1) > gpurun_out/tail_pmc_summary.json 2>/dev/null; echo "pmc summary exit $?"
head -11 gpurun_out/tail_kernel_stats_dropin.txt | cut -c1-150
