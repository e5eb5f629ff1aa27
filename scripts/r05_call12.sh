#!/bin/bash
# K3 shares out the emission of heavy blocks: tests, the metric frame, the train_post-shaped extra
set -u
export TMPDIR=/tmp
for f in tests/test_raster_gpu.py tests/test_product_paths_gpu.py tests/test_lod_gpu.py; do timeout 600 python -m pytest $f -q -m gpu -rf --tb=short -p no:cacheprovider 2>&1 | tail -4; done
timeout 300 python scripts/bench_min.py 2 --no-secondary 2>&1 | tail -2
timeout 400 python bench.py --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', round(d['value'],1))
for k, v in d['extra'].items(): print(k, round(v.get('value',0),1), {kk: round(x,4) for kk,x in (v.get('stages_ms') or {}).items()})
"
echo "--- HGS_K3_SHARE=0"
HGS_K3_SHARE=0 timeout 400 python bench.py --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', round(d['value'],1))
for k, v in d['extra'].items(): print(k, round(v.get('value',0),1), {kk: round(x,4) for kk,x in (v.get('stages_ms') or {}).items()})
"
