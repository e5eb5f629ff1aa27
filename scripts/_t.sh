cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_residency_gpu.py -m gpu -x -q -s -p no:cacheprovider 2>&1 | tail -4
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-stage-timing --extras config5_budgeted_6gb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['extra'].items(): print(k, round(v['value'],1), v['unit'], v['config']['rendered_tau_px'], v['config']['cuts_per_frame'])"
