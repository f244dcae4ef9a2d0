#!/bin/bash
# Runs ON THE GPU BOX: interleaved same-box comparison of SEVERAL values of one environment switch on the default bench workload.
#   tools/ab3_env.sh VAR REPEATS V1 V2 [V3 ...]   -> gpurun_out/ab/ab_<VAR>.txt
set -eo pipefail
VAR=$1; REP=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ab
mkdir -p "$OUT"
F=$OUT/ab_$VAR.txt
: > "$F"
for i in $(seq "$REP"); do
  for v in "$@"; do
    line=$(env "$VAR=$v" python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-dropin 2>/dev/null | tail -1)
    echo "$VAR=$v: $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); k=d.get("per_kernel_ms") or {}; print(d["value"], "cells/s", d["ms_per_step"], "ms", {x: k[x] for x in ("gemm_qkv","gemm_fc1","gemm_fc2","gemm_proj","cell_qkv_attention","layernorm") if x in k})')" | tee -a "$F"
  done
done
