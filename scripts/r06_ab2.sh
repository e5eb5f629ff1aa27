#!/bin/bash
# A/B of library variants on the headline + extras, preceded by the raster / property tests on the PRODUCT library.
#   r06_ab2.sh "<pytest -k expr or 'none'>" name...       (X=extras list, default trained_like_10m)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
K="$1"; shift
if [ "$K" != none ]; then
  timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_properties_gpu.py tests/test_product_paths_gpu.py -q -m gpu -x -p no:cacheprovider -k "$K" 2>&1 | tail -5
fi
L=hierarchical-3d-gaussians_amd/lib/libhgs.so
cp $L /tmp/libhgs_product.so
run() {
  timeout 300 python bench.py --no-cpu-baseline --no-live-pmc --no-secondary --extras ${X:-trained_like_10m} 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(' metric', round(d['value'], 1), {k: round(v, 4) for k, v in (d.get('stages_ms') or {}).items()})
for k, v in (d.get('extra') or {}).items():
    print(' ', k, round(v.get('value', 0), 1), {kk: round(vv, 4) for kk, vv in (v.get('stages_ms') or {}).items()})
"
}
for rep in 1 2; do
  echo "=== product (run $rep)"; run
  for name in "$@"; do
    echo "=== variant $name (run $rep)"
    cp ab_variants/libhgs_$name.so $L
    run
    cp /tmp/libhgs_product.so $L
  done
done
