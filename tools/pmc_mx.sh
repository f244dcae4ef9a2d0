#!/bin/bash
# Runs ON THE GPU BOX: SQ wait / busy counters and L2 hit counters of the MX residual GEMM alone (tools/bench_mx_only.py), separate --pmc passes
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_mx
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
P3="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -o p$i -- python3 "$ROOT/tools/bench_mx_only.py" 1024 > "$OUT/p$i.txt" 2> "$OUT/p$i.log" || echo "pass $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
sums = defaultdict(lambda: defaultdict(float))
for path in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    pas = os.path.relpath(path, out).split(os.sep)[0]
    for row in csv.DictReader(open(path, newline="")):
        k = row["Kernel_Name"].split("(")[0][:60] + " grid " + row.get("Grid_Size", "?")
        c = row["Counter_Name"]
        sums[k][c if c != "GRBM_GUI_ACTIVE" else c + "@" + pas] += float(row["Counter_Value"])
with open(os.path.join(out, "summary.txt"), "w") as f:
    for k, s in sorted(sums.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE@p1", 0)):
        if "gemm_mx" not in k: continue
        g1 = s.get("GRBM_GUI_ACTIVE@p1", 0) / 8.0; wc = max(s.get("SQ_WAVE_CYCLES", 0), 1.0)
        line = (f"{k}: cycles {g1:.3e} mfma_busy {s.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*max(g1,1)):.3f} lds_active {s.get('SQ_LDS_IDX_ACTIVE',0)/(256*max(g1,1)):.3f} "
                f"bank_conf {s.get('SQ_LDS_BANK_CONFLICT',0)/(256*max(g1,1)):.4f} wait_any {s.get('SQ_WAIT_ANY',0)/wc:.3f} wait_inst {s.get('SQ_WAIT_INST_ANY',0)/wc:.3f} "
                f"wait_lds {s.get('SQ_WAIT_INST_LDS',0)/wc:.3f} active_any {s.get('SQ_ACTIVE_INST_ANY',0)/wc:.3f} active_valu {s.get('SQ_ACTIVE_INST_VALU',0)/wc:.3f} "
                f"active_vmem {s.get('SQ_ACTIVE_INST_VMEM',0)/wc:.3f} active_lds {s.get('SQ_ACTIVE_INST_LDS',0)/wc:.3f} "
                f"l2_hit {s.get('TCC_HIT_sum',0)/max(s.get('TCC_HIT_sum',0)+s.get('TCC_MISS_sum',0),1):.3f} (hits {s.get('TCC_HIT_sum',0):.3e} misses {s.get('TCC_MISS_sum',0):.3e})")
        print(line); f.write(line + "\n")
PY
find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*counter_collection.csv' -delete; find "$OUT" -name '*agent_info.csv' -delete
