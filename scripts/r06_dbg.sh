cd $GRAFT_REPO_ROOT
HGS_SORT_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc --no-secondary --no-stage-timing --extras trained_like_10m 2>&1 | grep "hgs\]" | sort | uniq -c | sort -rn | head -20
