#!/bin/bash
# Runs ON THE GPU BOX: default bench workload (one warm-up + two timed passes) over chunk sizes and segment-stream counts, one box,
# so the lines are comparable.  Output: gpurun_out/sweep/chunk_stream_sweep.txt
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sweep
mkdir -p "$OUT"
: > "$OUT/chunk_stream_sweep.txt"
for cfg in "1024 3" "512 3" "512 6" "768 3" "768 4" "1024 2" "1024 4" "1536 3" "2048 3" "1024 3"; do
  set -- $cfg
  line=$(python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin --chunk "$1" --streams "$2" 2>/dev/null | tail -1)
  echo "chunk $1 streams $2: $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "cells/s", d["ms_per_step"], "ms")')" | tee -a "$OUT/chunk_stream_sweep.txt"
done
