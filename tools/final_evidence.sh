set -eo pipefail
mkdir -p gpurun_out/r4ev
RIBCA_TEST_REPORT=1 python -m pytest tests -x -q -m gpu > gpurun_out/r4ev/gpu_tests.log 2>&1
tail -2 gpurun_out/r4ev/gpu_tests.log
bash tools/collect_profiles.sh r4 > gpurun_out/prof_r4_collect.log 2>&1
bash tools/collect_pmc_sq.sh r4 > gpurun_out/pmc_r4_collect.log 2>&1
tail -2 gpurun_out/pmc_r4_collect.log
