#!/bin/bash
# Runs ON THE GPU BOX: the round's evidence on the final sources -> gpurun_out/<tag>ev, gpurun_out/prof_<tag>, gpurun_out/pmc_<tag>
#   tools/final_evidence.sh r6   (then copy the summaries into profiles/r6/)
set -eo pipefail
TAG=${1:-r6}
mkdir -p gpurun_out/${TAG}ev
RIBCA_TEST_REPORT=1 python -m pytest tests -q -m gpu > gpurun_out/${TAG}ev/gpu_tests.log 2>&1 || true
tail -2 gpurun_out/${TAG}ev/gpu_tests.log
bash tools/collect_profiles.sh $TAG > gpurun_out/prof_${TAG}_collect.log 2>&1
bash tools/collect_pmc_sq.sh $TAG > gpurun_out/pmc_${TAG}_collect.log 2>&1
tail -2 gpurun_out/pmc_${TAG}_collect.log
