#!/bin/bash
# A/B of library variants (ab_variants/libhgs_<name>.so) on the trained-scale extra + headline; usage: r06_ab.sh name...
cd $GRAFT_REPO_ROOT
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
echo "=== product"; run
for name in "$@"; do
  echo "=== variant $name"
  cp ab_variants/libhgs_$name.so $L
  run
done
cp /tmp/libhgs_product.so $L
