cd $GRAFT_REPO_ROOT
L=hierarchical-3d-gaussians_amd/lib/libhgs.so; cp $L /tmp/prod.so
for v in product k8pad product; do
  if [ $v != product ]; then cp ab_variants/libhgs_$v.so $L; else cp /tmp/prod.so $L; fi
  echo "== $v"; rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o r -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary > /dev/null 2>&1
  python scripts/rocprof_summary.py $(ls /tmp/prof_$v/*.db | head -1) 2>/dev/null | grep "preprocess_bwd\|sh_bwd"
  rm -rf /tmp/prof_$v
done
cp /tmp/prod.so $L
