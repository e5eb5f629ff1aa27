cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
HGS_BENCH_DIAG=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-stage-timing --extras config5_50m_4k_render,config3_train_post 2>&1 | grep -v "diag frame\|diag host\|amdgpu.ids" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:round(v['value'],1) for k,v in d['extra'].items()})
    else: print(l.strip()[:160])"
done
