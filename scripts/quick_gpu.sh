#!/bin/bash
# Quick GPU check while tuning a kernel: the raster + scale parity tests, then the bench line reduced to the numbers
# that matter (frames/s of both schedules, per-stage milliseconds).  Usage: gpurun -- bash scripts/quick_gpu.sh [notests]
if [ "${1:-}" != "notests" ]; then
  timeout 600 python -m pytest tests/test_raster_gpu.py tests/test_scale_parity_gpu.py tests/test_properties_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -4
fi
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dropin', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 4), 'ms; batched', round((d.get('batched') or {}).get('value', 0), 1))
print({k: round(v, 4) for k, v in (d.get('stages_ms') or {}).items()})
"
