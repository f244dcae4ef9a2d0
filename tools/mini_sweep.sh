#!/bin/bash
# Runs ON THE GPU BOX: a short chunk / segment-stream re-check around the default (one box, comparable lines) -> gpurun_out/sweep/mini_sweep.txt
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sweep
mkdir -p "$OUT"
: > "$OUT/mini_sweep.txt"
for cfg in "1024 3" "1024 2" "1024 4" "768 4" "1536 3" "1024 3"; do
  set -- $cfg
  line=$(python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin --chunk "$1" --streams "$2" 2>/dev/null | tail -1)
  echo "chunk $1 streams $2: $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "cells/s", d["ms_per_step"], "ms")')" | tee -a "$OUT/mini_sweep.txt"
done
