#!/bin/bash
# Runs ON THE GPU BOX: per-kernel statistics (rocprofv3 --kernel-trace --stats, one stream, one warm-up + one timed pass of config 3) of the
# library under two values of an environment switch:   bash tools/ab_env_prof.sh RIBCA_PROJ_MX 0 1   -> gpurun_out/ab_<VAR>/trace_<value>/
set -eo pipefail
VAR=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ab_$VAR
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export $VAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$v" -o trace -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin --streams 1 > "$OUT/line_$v.json" 2> "$OUT/trace_$v.log"
  find "$OUT/trace_$v" -name '*kernel_trace.csv' -delete
  find "$OUT/trace_$v" -name '*agent_info.csv' -delete
done
du -sh "$OUT"
