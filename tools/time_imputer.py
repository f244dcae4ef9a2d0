#!/usr/bin/env python3
"""BASELINE config 5's extra stage on one GPU: marker imputation (MAE, encoder 768/12 on the present channel tiles, decoder 512/8 on all
of them) for ~100 k cells of the full 15-marker panel with one marker missing, then the five classifiers on the imputed patches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from multiplexed_image_annotator_amd import _lib, ops, synth

dev = _lib.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = 15
seed = synth.SEED_BASE + 5
mae = ops.MaeModel(synth.make_mae_state_dict("immune_full", seed, 12, 8), dev)
patches = (torch.rand((n, L, 40, 40), device=dev) * 2 - 1)
present = list(range(L - 1))                      # the last marker is missing (index list [0..13, -1])
for chunk in (1024, 2048, 4096):
    mae.impute(patches[:chunk * 2].clone(), present, chunk_cells=chunk)
    torch.cuda.synchronize()
    x = patches.clone()
    t0 = time.perf_counter()
    mae.impute(x, present, chunk_cells=chunk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"impute {n} cells, 1 of {L} markers missing, chunk {chunk}: {dt:.3f} s = {n / dt:.0f} cells/s ({3.4407 * n / dt / 1e3:.1f} TFLOP/s algorithmic)")
