#!/bin/bash
# SQ counters + kernel times of the trained-scale frame (args: passed to scripts/trained_loop.py after the frame count)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/scripts/trained_loop.py 6 "$@" > /tmp/kt.log 2>&1; echo "trace exit $?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc1 -o pmc -- python $R/scripts/trained_loop.py 3 "$@" > /tmp/pmc1.log 2>&1; echo "pmc1 exit $?"
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --kernel-trace -d /tmp/pmc2 -o pmc -- python $R/scripts/trained_loop.py 3 "$@" > /tmp/pmc2.log 2>&1; echo "pmc2 exit $?"
cd $R
python scripts/rocprof_summary.py $(ls /tmp/kt/*.db | head -1) > gpurun_out/r06_pmc_kernel_stats.txt 2>/dev/null
head -16 gpurun_out/r06_pmc_kernel_stats.txt | cut -c1-160
python scripts/pmc_summary.py A=$(ls /tmp/pmc1/*.db | head -1) B=$(ls /tmp/pmc2/*.db | head -1) > gpurun_out/r06_pmc.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_pmc.json"))
for k, c in d.items():
    if not isinstance(c, dict) or "SQ_WAVES" not in c: continue
    w = c["SQ_WAVES"]
    if w < 1: continue
    print(k[-48:], "waves", int(w), {x.replace("SQ_", ""): round(c[x] / w, 1) for x in sorted(c) if x != "SQ_WAVES"})
PY
