cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "lod or hier or raster" 2>&1 | tail -2
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --extras config5_50m_4k_render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['extra'].items(): print(k, round(v['value'],1), v['unit'], {a:round(b,3) for a,b in v['stages_ms'].items()})
print('headline', round(d['value'],1), {a:round(b,4) for a,b in d['stages_ms'].items()})"
bash scripts/prof_config5.sh 2>&1 | grep "lod_\|scan_block"
