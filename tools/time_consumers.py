#!/usr/bin/env python3
"""Times the post-predict consumers on the GPU at the BASELINE config-3 size (4096 x 4096 mask, ~100 k cells): label painting
(ribca_colorize) and the 25-nearest-neighbour co-occurrence (ribca_knn_cooccurrence)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from multiplexed_image_annotator_amd import _lib, colors, ops, synth

dev = _lib.require_gpu()
mask, _ = synth.make_mask_and_image(4096, 4096, 100000, 1, synth.SEED_BASE + 3, device=dev, want_image=False)
mask = mask.to(torch.int32)
ids, tab = ops.label_table(mask)
n = len(ids)
rng = np.random.default_rng(0)
tidx = rng.integers(0, 12, n)
conf = rng.random(n).astype(np.float32)
pal = np.array(colors.get_colors(12), np.uint8)
x = tab[:, 5] / tab[:, 6]
y = tab[:, 4] / tab[:, 6]
for name, fn in (("colorize", lambda: ops.colorize(mask, ids, pal[tidx], colors.confidence_colors(conf), (tidx + 1).astype(np.uint8))),
                 ("knn25", lambda: ops.knn_cooccurrence(x, y, tidx, 12, 25)),
                 ("compositions (201-NN, 8 sizes)", lambda: ops.knn_compositions(x, y, tidx, 12))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0):.1f} ms for {n} cells, 4096x4096 (host table upload included)")
