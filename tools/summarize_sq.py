#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/collect_pmc_sq.sh to per-kernel counter sums and the derived duty figures
DESIGN.md quotes.  Normalisation (MI355X_MICROARCH.md, cycle-constants table): GRBM_GUI_ACTIVE is summed over the 8 XCDs, so
kernel cycles = GRBM_GUI_ACTIVE / 8; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, so full duty = 1024 SIMDs x kernel cycles;
SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT are LDS-array cycles per CU (full duty = 256 CUs x kernel cycles); SQ_WAVE_CYCLES,
SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles per wave and are reported as fractions of SQ_WAVE_CYCLES."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
sums = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
for path in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    pas = os.path.relpath(path, out).split(os.sep)[0]
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0].replace("void ribca::", "").replace("ribca::", "")
            c = row["Counter_Name"]
            key = c if c != "GRBM_GUI_ACTIVE" else c + "@" + pas
            sums[k][key] += float(row["Counter_Value"])
            calls[k][pas].add(row.get("Dispatch_Id"))
res = {}
lines = []
for k, s in sorted(sums.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE@p1", 0.0)):
    gui1 = s.get("GRBM_GUI_ACTIVE@p1", 0.0) / 8.0
    gui2 = s.get("GRBM_GUI_ACTIVE@p2", 0.0) / 8.0
    if gui1 <= 0:
        continue
    wc = max(s.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    d = {"dispatches": max(len(v) for v in calls[k].values()), "kernel_cycles": gui1,
         "mfma_busy_frac": s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui1),
         "lds_active_frac": s.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * gui1),
         "lds_bank_conflict_frac": s.get("SQ_LDS_BANK_CONFLICT", 0.0) / (256.0 * gui1),
         "sq_busy_frac": s.get("SQ_BUSY_CYCLES", 0.0) / (8.0 * gui1) if s.get("SQ_BUSY_CYCLES") else None,
         "wait_any_of_wave_cycles": s.get("SQ_WAIT_ANY", 0.0) / wc,
         "wait_inst_any_of_wave_cycles": s.get("SQ_WAIT_INST_ANY", 0.0) / wc,
         "wait_inst_lds_of_wave_cycles": s.get("SQ_WAIT_INST_LDS", 0.0) / wc,
         "raw": {c: v for c, v in s.items()}}
    if gui2 > 0:
        d["mfma_valu_coexec_frac"] = s.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / (1024.0 * gui2)
        d["insts_mfma_per_kcycle_per_simd"] = s.get("SQ_INSTS_MFMA", 0.0) / (1024.0 * gui2) * 1000.0
    res[k] = d
    lines.append(f"{k[:70]:70s} n={d['dispatches']:5d} mfma_busy {d['mfma_busy_frac']:.3f}  lds_active {d['lds_active_frac']:.3f}  "
                 f"bank_conf {d['lds_bank_conflict_frac']:.4f}  wait_any {d['wait_any_of_wave_cycles']:.3f}  wait_inst {d['wait_inst_any_of_wave_cycles']:.3f}  "
                 f"wait_lds {d['wait_inst_lds_of_wave_cycles']:.3f}")
import hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from multiplexed_image_annotator_amd import build as _build
    res["kernel_source_sha256"] = _build.source_fingerprint()
except Exception:
    res["kernel_source_sha256"] = None
json.dump(res, open(os.path.join(out, "sq_summary.json"), "w"), indent=1)
open(os.path.join(out, "sq_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:25]))
