cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_scale_parity_gpu.py tests/test_product_paths_gpu.py -m gpu -x -q -p no:cacheprovider -k "crowded or 4k or heavy or trained" 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --extras config5_budgeted_6gb,heavy_1m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['extra'].items(): print(k, round(v['value'],1), v['unit'], {a:round(b,3) for a,b in v.get('stages_ms',{}).items()})
print('headline', round(d['value'],1), d['stages_ms']['tile_depth_sort'])"
