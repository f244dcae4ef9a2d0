#!/bin/bash
# Runs ON THE GPU BOX (through gpurun) and leaves the round's profiling evidence under gpurun_out/prof_<tag>/:
#   1. rocprofv3 --kernel-trace --stats of the default bench workload (one warm-up + one timed step) on ONE stream, so that a
#      kernel's traced duration is its own (bench.py's roofline pass measures the same way)   -> kernel_stats.csv
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) on a reduced tile with full-size GEMM launches
#      (1024-cell chunks)                                                                   -> pmc_*.csv
# tools/summarize_pmc.py then turns (2) into profiles/<tag>/gemm_traffic.json, which bench.py reads for roofline.traffic.
set -eo pipefail
TAG=${1:-r1_final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[prof] kernel trace"; date
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin --streams 1 > "$OUT/bench_line_under_rocprof.json" 2> "$OUT/trace.log"
echo "[prof] FETCH_SIZE"; date
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-dropin --cells 8000 --size 1280 > "$OUT/fetch.json" 2> "$OUT/fetch.log"
echo "[prof] WRITE_SIZE"; date
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-dropin --cells 8000 --size 1280 > "$OUT/write.json" 2> "$OUT/write.log"
echo "[prof] summarising"; date
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" "$TAG"
# keep what travels back small: the per-dispatch traces are hundreds of MB
find "$OUT" -name '*kernel_trace.csv' -delete
find "$OUT" -name '*counter_collection.csv' -delete
find "$OUT" -name '*agent_info.csv' -delete
du -sh "$OUT"
