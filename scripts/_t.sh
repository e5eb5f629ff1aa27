cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_residency_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-stage-timing --extras config5_budgeted_6gb 2>&1 | grep -v amdgpu | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps(d['extra'],indent=1))
    else: print(l.rstrip()[:300])"
