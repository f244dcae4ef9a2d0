#!/bin/bash
# Runs ON THE GPU BOX: the vector-memory path of the MX residual GEMM alone (tools/bench_mx_only.py: fc2 at D = 576 / 384 / 288, M = 103 424) seen by
# the TA / TCP / SQ counters, for the product build and the no-A / no-W timing ablations (libribca_ab_noa.so / libribca_ab_now.so, built in the
# container with tools/build_ab_lib.py noa WORK -DMXDBG_NOA etc.).  Separate --pmc passes, the program directly behind `--`, no trace domain
# besides --kernel-trace.   ->  gpurun_out/pmc_ta/summary.txt  (copy into profiles/<round>/)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_ta
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/list_avail.txt" 2>&1 || true
grep -oE "\b(TA|TCP|TD|SQ|TCC)_[A-Za-z0-9_]+" "$OUT/list_avail.txt" | sort -u > "$OUT/counter_names.txt"
pick() { for c in "$@"; do grep -qx "$c" "$OUT/counter_names.txt" && printf '%s ' "$c"; done; }
# (two counters of one block per pass: more "exceeds the capabilities of the hardware" on gfx950 for TA / TCP)
PASSES=(
  "$(pick TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum)"
  "$(pick TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum)"
  "$(pick SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES)"
  "$(pick TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum)"
  "$(pick TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum)"
  "$(pick TCC_HIT_sum TCC_MISS_sum)"
)
for LIB in libribca_hip.so libribca_ab_noa.so libribca_ab_now.so; do
  [ -f "$ROOT/multiplexed-image-annotator_amd/$LIB" ] || { echo "missing $LIB"; continue; }
  tag=${LIB%.so}
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1))
    [ -z "$P" ] && continue
    echo "[$tag] pass $i: $P"
    RIBCA_LIB=$LIB timeout -k 10 200 rocprofv3 --pmc $P GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/$tag/p$i" -o p$i -- python3 "$ROOT/tools/bench_mx_only.py" 1024 $tag \
      > "$OUT/$tag.p$i.txt" 2> "$OUT/$tag.p$i.log" || echo "$tag pass $i failed"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
lines = []
for tagdir in sorted(glob.glob(os.path.join(out, "libribca_*"))):
    if not os.path.isdir(tagdir): continue
    sums = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(tagdir, "p*", "**", "*counter_collection.csv"), recursive=True):
        pas = os.path.relpath(path, tagdir).split(os.sep)[0]
        for row in csv.DictReader(open(path, newline="")):
            if "gemm_mx" not in row["Kernel_Name"]: continue
            k = "grid " + row.get("Grid_Size", "?")
            c = row["Counter_Name"]
            c = c if c != "GRBM_GUI_ACTIVE" else c + "@" + pas
            sums[k][c] += float(row["Counter_Value"]); n[k][c] += 1
    for k, s in sorted(sums.items()):
        per = {c: v / max(n[k][c], 1) for c, v in s.items()}      # per launch
        g = per.get("GRBM_GUI_ACTIVE@p3", per.get("GRBM_GUI_ACTIVE@p1", 0)) / 8.0      # summed over 8 XCDs
        lines.append(f"{os.path.basename(tagdir)} {k}: gpu cycles per launch {g:.4e}")
        for c in sorted(per):
            if c.startswith("GRBM"): continue
            v = per[c]
            extra = ""
            if c.startswith("TA_") or c.startswith("TCP_"):
                extra = f"   per CU-cycle {v / (256 * max(g, 1)):.4f}"
            elif c.startswith("SQ_"):
                extra = f"   per SIMD-cycle {v / (1024 * max(g, 1)):.4f}   per CU-cycle {v / (256 * max(g, 1)):.4f}"
            lines.append(f"    {c} = {v:.4e}{extra}")
with open(os.path.join(out, "summary.txt"), "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*counter_collection.csv' -delete; find "$OUT" -name '*agent_info.csv' -delete
rm -f "$OUT/list_avail.txt"
