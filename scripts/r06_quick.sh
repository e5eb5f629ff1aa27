#!/bin/bash
# Round 6 quick lease: selected tests (-k expression in $1, files in $2), then the trained-scale extras.
set -u
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
K="${1:-long_run or trained_scale or awkward}"; F="${2:-tests/test_raster_gpu.py tests/test_scale_parity_gpu.py}"; X="${3:-trained_like_10m,trained_cut_10m}"
timeout 1200 python -m pytest $F -q -m gpu -x -p no:cacheprovider -k "$K" 2>&1 | tail -12
[ "$X" = none ] && exit 0
timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --extras $X > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/quick_bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "batched", round((d.get("batched") or {}).get("value", 0), 1), "stages", {k: round(v, 4) for k, v in d["stages_ms"].items()})
for k, v in (d.get("extra") or {}).items():
    print(" extra", k, v.get("error") or (round(v.get("value", 0), 1), v.get("unit"), "ms", round(v.get("ms_per_step", 0), 3), "host", round(v.get("host_ms_per_step") or 0, 3), "miss", v["config"].get("capacity_misses")))
    print("   stages", {kk: round(vv, 4) for kk, vv in (v.get("stages_ms") or {}).items()}, "sum", round(v.get("stage_sum_ms") or 0, 4))
PY
