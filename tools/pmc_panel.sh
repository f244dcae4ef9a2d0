#!/bin/bash
# Runs ON THE GPU BOX: the W-panel tile walk of the MX kernel (RIBCA_MX_PANEL_KB) on immune_full's launches -- time per
# kernel class without the profiler, then L2 hit / miss, L2 read latency as the vector L1 sees it, and FETCH_SIZE per MX kernel with it.
#   -> gpurun_out/pmc_panel/summary.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_panel
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/timing.txt"
for rep in 1 2; do for KB in 0 2048 1024; do
  RIBCA_MX_PANEL_KB=$KB python3 "$ROOT/tools/bench_block.py" immune_full 2>/dev/null | sed "s/^/panel_kb=$KB /" >> "$OUT/timing.txt"
done; done
for KB in 0 2048; do
  i=0
  for P in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE"; do
    i=$((i+1))
    RIBCA_MX_PANEL_KB=$KB RIBCA_BENCH_CELLS=4096 timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/kb$KB/p$i" -o p$i -- python3 "$ROOT/tools/bench_block.py" immune_full \
      > "$OUT/kb$KB.p$i.txt" 2> "$OUT/kb$KB.p$i.log" || echo "kb$KB pass $i failed"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
lines = [l.rstrip() for l in open(os.path.join(out, "timing.txt"))]
for kb in ("kb0", "kb2048"):
    sums = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(out, kb, "p*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            name = row["Kernel_Name"]
            if "gemm_mx_duo" not in name: continue
            m = re.search(r"Epi\w+", name)
            k = m.group(0) if m else name[:40]
            sums[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
    for k, s in sorted(sums.items()):
        per = {c: v / max(n[k][c], 1) for c, v in s.items()}
        hit, miss = per.get("TCC_HIT_sum", 0), per.get("TCC_MISS_sum", 0)
        lat = per.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(per.get("TCP_TCC_READ_REQ_sum", 1), 1)
        lines.append(f"{kb} gemm_mx_duo<{k}>: launches {max(n[k].values())}, L2 hit rate {hit / max(hit + miss, 1):.3f} (hits {hit:.3e} misses {miss:.3e} per launch), "
                     f"mean L2 read latency seen by the vector L1 {lat:.0f} cycles, 2 x FETCH_SIZE {2 * per.get('FETCH_SIZE', 0) / 1e6:.1f} MB per launch")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*counter_collection.csv' -delete; find "$OUT" -name '*agent_info.csv' -delete
