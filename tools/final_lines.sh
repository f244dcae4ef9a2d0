#!/bin/bash
# Runs ON THE GPU BOX: the bench lines and diagnostic-tool outputs committed under profiles/<round>/ (gpurun_out/r3fin/ here).
# Stage 1 = bench lines, stage 2 = diagnostic tools (needs build --diag).
set -eo pipefail
mkdir -p gpurun_out/r3fin
if [ "${1:-all}" != "tools" ]; then
python bench.py > gpurun_out/r3fin/bench_default.json 2> gpurun_out/r3fin/bench_default.log
python bench.py --impute --no-cpu-baseline > gpurun_out/r3fin/bench_impute.json 2> gpurun_out/r3fin/bench_impute.log
python bench.py --config1 --no-cpu-baseline > gpurun_out/r3fin/bench_config1.json 2> gpurun_out/r3fin/bench_config1.log
fi
RIBCA_SHARE_GPU=1 RIBCA_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 --cells 20000 --size 2048 --no-cpu-baseline --no-roofline --no-dropin > gpurun_out/r3fin/bench_gpus2_gloo_shared_gpu.json 2> gpurun_out/r3fin/bench_gpus2_gloo_shared_gpu.log
python tools/stamp_gemm.py > gpurun_out/r3fin/stamp_gemm.txt 2>&1
python tools/stamp_duo.py > gpurun_out/r3fin/stamp_duo.txt 2>&1
python tools/bench_cell_attention.py > gpurun_out/r3fin/bench_cell_attention.txt 2>&1
