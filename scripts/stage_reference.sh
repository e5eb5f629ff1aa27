#!/bin/bash
# Stage the reference's Python tree for ONE gpurun lease (VERDICT r03 "do this" 1).
#
# The GPU boxes carry no checkout of the reference, so tests/test_reference_on_gpu.py (the acceptance sentence: the
# reference's UNMODIFIED scripts on the HIP op) skips there.  gpurun ships the working tree including untracked files,
# so the .py files the three scripts import are placed in .refstage/ -- gitignored, never committed, removed again by
# `stage_reference.sh clean` right after the lease -- and HGS_REFERENCE points the tests at it:
#
#   scripts/stage_reference.sh stage
#   gpurun -- 'HGS_REFERENCE=$PWD/.refstage python -m pytest tests/test_reference_on_gpu.py -m gpu -rA ...'
#   scripts/stage_reference.sh clean
#
# Only what train_single.py / train_post.py / render_hierarchy.py import is staged (no viewer, no preprocessing, no
# assets).  Nothing in the product reads .refstage/.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF="${HGS_REFERENCE_SRC:-/root/reference}"
STAGE="$ROOT/.refstage"
case "${1:-stage}" in
  stage)
    rm -rf "$STAGE"
    mkdir -p "$STAGE"
    for f in train_single.py train_post.py render_hierarchy.py train_coarse.py; do cp "$REF/$f" "$STAGE/"; done
    for d in arguments scene gaussian_renderer utils lpipsPyTorch; do
      (cd "$REF" && find "$d" -name '*.py' -exec cp --parents {} "$STAGE/" \;)
    done
    grep -qx '.refstage/' "$ROOT/.gitignore" || { echo "refusing: .refstage/ is not gitignored" >&2; rm -rf "$STAGE"; exit 1; }
    echo "staged $(find "$STAGE" -name '*.py' | wc -l) files into $STAGE"
    ;;
  clean)
    rm -rf "$STAGE"
    echo "removed $STAGE"
    ;;
  *) echo "usage: $0 stage|clean" >&2; exit 2;;
esac
