#!/usr/bin/env python3
"""Run one classifier twice on the same patches with different chunkings / stream counts and report where the probabilities differ
bit for bit (they must not: tests/test_gpu_e2e.py::test_config3_full_size_properties).  usage: check_determinism.py [model] [cells]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from multiplexed_image_annotator_amd import _lib, ops, synth

dev = _lib.require_gpu()
name = sys.argv[1] if len(sys.argv) > 1 else "nerve"
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
d, c, k = synth.VIT_CONFIGS[name]
g = torch.Generator().manual_seed(5)
patches = (torch.rand((cells, c, 40, 40), generator=g) * 2 - 1).to(dev)
model = ops.VitModel(synth.make_vit_state_dict(name, synth.SEED_BASE + 3), dev)
src = list(range(c))
ref = model.predict_proba(patches, src, chunk_cells=1024, streams=3)
for chunk, streams in ((1024, 3), (1024, 1), (300, 1), (300, 1), (777, 2), (1024, 3)):
    p = model.predict_proba(patches, src, chunk_cells=chunk, streams=streams)
    bad = (p != ref).any(dim=1).nonzero().flatten()
    print(f"{name} chunk {chunk} streams {streams}: {len(bad)} of {cells} rows differ" + (f", first {bad[:8].tolist()}, max |d| {(p - ref).abs().max().item():.3e}" if len(bad) else ""), flush=True)
