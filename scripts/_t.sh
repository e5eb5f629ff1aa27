cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "lod or hier or config3" 2>&1 | tail -2
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --extras config5_50m_4k_render,config3_train_post 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['extra'].items(): print(k, round(v['value'],1), v['unit'], {a:round(b,3) for a,b in v['stages_ms'].items()})
print('headline', round(d['value'],1))"
python scripts/bench_next.py 2>/dev/null | grep "f-1\|train_post"
