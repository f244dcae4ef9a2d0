#!/usr/bin/env python3
"""Reduce the rocprofv3 outputs of tools/collect_profiles.sh to the small files committed under profiles/<tag>/:
kernel_stats.csv (per-kernel totals of the traced bench step) and gemm_traffic.json (HBM-side bytes per GEMM launch from the
FETCH_SIZE / WRITE_SIZE passes, with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE tallies 128-byte requests at
64 bytes, so it is doubled; both counters are in KiB... the guide's unit is KB = 1024 bytes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def find(sub, suffix):
    hits = glob.glob(os.path.join(out, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def counter_sums(sub, counter):
    path = find(sub, "counter_collection.csv")
    per_kernel = defaultdict(lambda: [0.0, 0])
    if path is None:
        return per_kernel
    seen = set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            per_kernel[name][0] += float(row["Counter_Value"])
            key = (row.get("Dispatch_Id"), name)
            if key not in seen:
                seen.add(key)
                per_kernel[name][1] += 1
    return per_kernel


def counter_by_shape(sub, counter):
    """KiB and launches per (kernel, grid size): one kernel instantiation serves several GEMM shapes"""
    path = find(sub, "counter_collection.csv")
    acc = defaultdict(lambda: [0.0, set()])
    if path is None:
        return acc
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            key = (row["Kernel_Name"], int(row.get("Grid_Size", 0) or 0))
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1].add(row.get("Dispatch_Id"))
    return acc


fetch = counter_sums("fetch", "FETCH_SIZE")
write = counter_sums("write", "WRITE_SIZE")
gemm = [k for k in fetch if "gemm_ps_split_kernel" in k or "gemm_ps_duo_kernel" in k or "gemm_mx_duo_kernel" in k or "cell_qkv_attention_kernel" in k]
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 "
                 "--no-cpu-baseline --no-roofline --cells 8000 --size 1280 (chunk 1024)",
       "kernel_family": "gemm_mx_duo_kernel + gemm_ps_duo_kernel + gemm_ps_split_kernel + cell_qkv_attention_kernel (all instantiations)",
       "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; "
                     "unit KB = 1024 B",
       "note": "counts L2 misses served by the Infinity Cache as well as HBM; per-launch figure (M = 103424 rows for full chunks)"}
launches = sum(fetch[k][1] for k in gemm)
f_kb = sum(fetch[k][0] for k in gemm)
w_kb = sum(write[k][0] for k in gemm if k in write)
res["launches"] = launches
res["fetch_size_kb_sum"] = f_kb
res["write_size_kb_sum"] = w_kb
if launches:
    res["traffic_bytes_per_launch"] = (2.0 * f_kb + w_kb) * 1024.0 / launches
    res["per_kernel_bytes_per_launch"] = {
        k.split("(")[0].replace("void ribca::", ""): (2.0 * fetch[k][0] + write.get(k, [0.0, 0])[0]) * 1024.0 / max(fetch[k][1], 1) for k in gemm}
    other = {}
    for k in fetch:
        if k in gemm or fetch[k][1] == 0:
            continue
        other[k.split("(")[0].replace("void ribca::", "")[:90]] = {
            "launches": fetch[k][1], "bytes_per_launch": (2.0 * fetch[k][0] + write.get(k, [0.0, 0])[0]) * 1024.0 / fetch[k][1]}
    res["other_kernels"] = other
fs, ws = counter_by_shape("fetch", "FETCH_SIZE"), counter_by_shape("write", "WRITE_SIZE")
res["per_shape"] = [
    {"kernel": k.split("(")[0].replace("void ribca::", "")[:90], "grid_threads": grid, "launches": len(v[1]),
     "fetch_bytes_per_launch": 2.0 * v[0] * 1024.0 / max(len(v[1]), 1),
     "write_bytes_per_launch": ws[(k, grid)][0] * 1024.0 / max(len(ws[(k, grid)][1]), 1) if (k, grid) in ws else None}
    for (k, grid), v in sorted(fs.items(), key=lambda kv: -kv[1][0]) if k in gemm][:40]
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
try:
    from multiplexed_image_annotator_amd import build as _build
    res["kernel_source_sha256"] = _build.source_fingerprint()
except Exception:
    res["kernel_source_sha256"] = None
# every kernel of the ViT forward (GEMMs, attention, statistics / LayerNorm, embed, head): counter bytes per CELL of the reduced pass
vit = [k for k in fetch if "ribca::" in k and any(t in k for t in ("gemm_ps_", "gemm_mx_", "mx_pack_act", "attention", "cell_qkv", "layernorm", "row_stats", "ln_finalize", "embed_f32", "head_softmax", "cls_rows"))]
n_cells_line = None
try:
    n_cells_line = json.load(open(os.path.join(out, "fetch.json")))["config"]["cells"]
except Exception:
    pass
if n_cells_line:
    vb = sum((2.0 * fetch[k][0] + write.get(k, [0.0, 0])[0]) * 1024.0 for k in vit)
    res["vit_bytes_per_cell"] = vb / n_cells_line
    res["vit_bytes_per_cell_by_kernel"] = {k.split("(")[0].replace("void ribca::", "")[:80]: (2.0 * fetch[k][0] + write.get(k, [0.0, 0])[0]) * 1024.0 / n_cells_line for k in vit}
    res["cells_in_counter_pass"] = n_cells_line
dst = os.path.join(root, "gpurun_out", "prof_" + tag)
os.makedirs(dst, exist_ok=True)
with open(os.path.join(dst, "gemm_traffic.json"), "w") as f:
    json.dump(res, f, indent=1)
stats = find("trace", "kernel_stats.csv")
if stats:
    with open(stats) as f, open(os.path.join(dst, "kernel_stats.csv"), "w") as g:
        g.write(f.read())
dom = find("trace", "domain_stats.csv")
if dom:
    with open(dom) as f, open(os.path.join(dst, "domain_stats.csv"), "w") as g:
        g.write(f.read())
print(json.dumps({k: res[k] for k in ("launches", "traffic_bytes_per_launch") if k in res}))
