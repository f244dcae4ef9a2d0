#!/bin/bash
# Runs ON THE GPU BOX: the forward / golden / repeatability / imputer tests under each A/B switch of the library (the non-default paths
# must stay green: they are what the A/B measurements compare against)  ->  gpurun_out/gpu_tests_under_switches.log
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/gpu_tests_under_switches.log
: > $OUT
for SW in RIBCA_CELL_ATTN=0 RIBCA_MXZ=0 RIBCA_MX=0 RIBCA_MAE_FOLD=0 RIBCA_MARGIN_PROBE=0; do
  echo "== $SW" >> $OUT
  env $SW python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -m gpu \
    -k "vit_forward or golden or repeatable or mae or config1 or config2 or two_ranks_match" 2>&1 | tail -2 >> $OUT
done
cat $OUT
