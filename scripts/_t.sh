cd $GRAFT_REPO_ROOT
L=hierarchical-3d-gaussians_amd/lib/libhgs.so; cp $L /tmp/prod.so
for v in product w5 b56 product; do
  if [ $v != product ]; then cp ab_variants/libhgs_$v.so $L; else cp /tmp/prod.so $L; fi
  echo "== $v"; python scripts/bench_min.py 2 --steps 40 --warmup 5 --no-secondary
done
cp /tmp/prod.so $L
