#!/bin/bash
# Tuning aid (GPU box): run the quick bench with each variant built by scripts/ab_build.sh; restores the product library.
L=hierarchical-3d-gaussians_amd/lib/libhgs.so
cp $L /tmp/libhgs_product.so
for name in "$@"; do
  echo "=== variant $name"
  cp ab_variants/libhgs_$name.so $L
  bash scripts/quick_gpu.sh notests
done
cp /tmp/libhgs_product.so $L
