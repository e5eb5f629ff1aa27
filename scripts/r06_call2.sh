#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --extras trained_like_10m,trained_cut_10m,config3_train_post > gpurun_out/r06_call2_bench.json 2> gpurun_out/r06_call2_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_call2_bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "stages", {k: round(v, 4) for k, v in d["stages_ms"].items()})
for k, v in (d.get("extra") or {}).items():
    print(" extra", k, v.get("error") or (round(v.get("value", 0), 1), v.get("unit"), "ms", round(v.get("ms_per_step", 0), 3), "host", v.get("host_ms_per_step"), v["config"]))
    print("   stages", {kk: round(vv, 4) for kk, vv in (v.get("stages_ms") or {}).items()}, "sum", v.get("stage_sum_ms"))
PY
timeout 600 python -m pytest tests/test_product_paths_gpu.py -q -m gpu -x -p no:cacheprovider -k "capacity" 2>&1 | tail -5
bash scripts/ab_run.sh strict 2>&1 | tail -4
