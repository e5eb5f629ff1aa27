set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=hierarchical-3d-gaussians_amd/lib/libhgs.so; cp $L /tmp/prod.so
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_properties_gpu.py tests/test_product_paths_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
for v in product thr1 product; do
  if [ $v != product ]; then cp ab_variants/libhgs_$v.so $L; else cp /tmp/prod.so $L; fi
  echo "== $v"; python scripts/bench_min.py 3 --steps 40 --warmup 5 --no-secondary
done
cp /tmp/prod.so $L
echo "== product, count by copy"; HGS_COUNT_BY_COPY=1 python scripts/bench_min.py 3 --steps 40 --warmup 5 --no-secondary
