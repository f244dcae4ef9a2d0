#!/usr/bin/env python3
"""cells/s of ONE classifier at several chunk sizes and stream counts (the tile grid of a narrow classifier's GEMMs is only 1-5 rounds of
workgroups at 1024 cells: does a larger chunk pay for it?).  usage: python tools/chunk_by_model.py [cells] [chunk,chunk,...]
(run with RIBCA_CHUNK_SCALE=1 for the raw chunk sizes: the 576-wide classifier otherwise runs at 4 x the chunk asked for)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multiplexed_image_annotator_amd import _lib, ops, synth

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
chunks = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1024, 1536, 2048, 3072, 4096)
dev = _lib.require_gpu()
for name in ("nerve", "immune_base", "immune_extended", "immune_full"):
    d, c, k = synth.VIT_CONFIGS[name]
    model = ops.VitModel(synth.make_vit_state_dict(name, 1), device=dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    patches = torch.randn((cells, c, 40, 40), generator=g).to(dev)
    src = list(range(c))
    row = []
    for chunk in chunks:
        for streams in (3,):
            model.predict_proba(patches, src, chunk_cells=chunk, streams=streams)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                model.predict_proba(patches, src, chunk_cells=chunk, streams=streams)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            row.append(f"chunk {chunk}: {cells / best:7.1f} k cells/s")
    print(f"{name:16s} D={d} {cells} cells, 3 streams | " + " | ".join(row), flush=True)
    del model, patches
