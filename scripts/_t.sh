cd $GRAFT_REPO_ROOT
L=hierarchical-3d-gaussians_amd/lib/libhgs.so; cp $L /tmp/prod.so
for v in product noclear product; do
  if [ $v != product ]; then cp ab_variants/libhgs_$v.so $L; else cp /tmp/prod.so $L; fi
  echo "== $v"; bash scripts/prof_config5.sh 2>&1 | grep "lod_mark\|lod_emit\|lod_weights"
done
cp /tmp/prod.so $L
