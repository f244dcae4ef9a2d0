#!/bin/bash
# Runs ON THE GPU BOX: LDS counters of the fused per-cell kernel and of the unfused attention kernels (tools/bench_cell_attention.py, 1024 cells per
# width) for the product library and, when present, alternative builds (tools/build_ab_lib.py).   tools/pmc_cell_lds.sh [libribca_ab_x.so ...]
#   ->  gpurun_out/pmc_cell_lds/summary.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_cell_lds
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for LIB in libribca_hip.so "$@"; do
  tag=${LIB%.so}
  RIBCA_DIAG=0 RIBCA_LIB=$LIB timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/$tag" -o p -- python3 "$ROOT/tools/bench_cell_attention.py" 1024 \
    > "$OUT/$tag.txt" 2> "$OUT/$tag.log" || echo "$tag failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
lines = []
for tagdir in sorted(glob.glob(os.path.join(out, "libribca_*"))):
    if not os.path.isdir(tagdir): continue
    s = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(tagdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            k = row["Kernel_Name"]
            if "attention" not in k: continue
            k = re.sub(r"\(.*", "", k).replace("void ribca::", "")
            s[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
    for k in sorted(s):
        per = {c: v / max(n[k][c], 1) for c, v in s[k].items()}
        act = max(per.get("SQ_LDS_IDX_ACTIVE", 0), 1)
        lines.append(f"{os.path.basename(tagdir):24s} {k:48s} gpu cycles {per.get('GRBM_GUI_ACTIVE', 0) / 8:.4e}  lds_idx_active {act:.4e}  bank_conflict {per.get('SQ_LDS_BANK_CONFLICT', 0):.4e} "
                     f"({per.get('SQ_LDS_BANK_CONFLICT', 0) / act:.3f} of active)  addr_conflict {per.get('SQ_LDS_ADDR_CONFLICT', 0):.3e}  unaligned {per.get('SQ_LDS_UNALIGNED_STALL', 0):.3e}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*counter_collection.csv' -delete; find "$OUT" -name '*agent_info.csv' -delete
