#!/usr/bin/env python3
"""per-kernel time of ONE classifier's forward (the library's own event profile): which of a block's launches pays under a switch
(RIBCA_MX / RIBCA_MXZ / RIBCA_CELL_ATTN are read once per process, so run one process per setting).
usage: python tools/bench_block.py [model names ...]   (default: immune_extended immune_full); cells via RIBCA_BENCH_CELLS (4096)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multiplexed_image_annotator_amd import _lib, ops, synth

names = sys.argv[1:] or ["immune_extended", "immune_full"]
cells = int(os.environ.get("RIBCA_BENCH_CELLS", "4096"))
dev = _lib.require_gpu()
tag = " ".join(f"{k}={os.environ[k]}" for k in ("RIBCA_MX", "RIBCA_MXZ", "RIBCA_CELL_ATTN") if k in os.environ) or "default"
for name in names:
    d, c, k = synth.VIT_CONFIGS[name]
    model = ops.VitModel(synth.make_vit_state_dict(name, 1), device=dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    patches = torch.randn((cells, c, 40, 40), generator=g).to(dev)
    src = list(range(c))
    model.predict_proba(patches, src, recheck=[])      # warm-up
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        ops.prof_enable(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model.predict_proba(patches, src, recheck=[])
        e1.record()
        torch.cuda.synchronize()
        pm = ops.prof_read()
        ops.prof_enable(False)
        tot = e0.elapsed_time(e1)
        if best is None or tot < best[0]:
            best = (tot, pm)
    tot, pm = best
    line = " ".join(f"{k2.replace('gemm_', '')}={v[0]:.1f}" for k2, v in pm.items() if v[1])
    print(f"[{tag}] {name} D={d} {cells} cells: {tot:.1f} ms ({cells / tot * 1e3:.0f} cells/s) | {line}", flush=True)
