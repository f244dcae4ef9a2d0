#!/usr/bin/env python3
"""Per-stage times of the pre-processing rows (P1-P7) and the vote at BASELINE config 3 (15-ch 4096 x 4096, ~100 k cells), with the
algorithmic bytes each stage has to move -> achieved GB/s against the ~8 TB/s HBM peak."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from multiplexed_image_annotator_amd import _lib, ops, synth

dev = _lib.require_gpu()
C, S, N = 15, 4096, 100000
mask, img = synth.make_mask_and_image(S, S, N, C, synth.SEED_BASE + 3, device=dev)
mask = mask.to(torch.int32)
raw = img.to(torch.int16)
del img


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


t_norm, image = timed(lambda: ops.normalize_image(raw, blur=0.3, amax=99.8, u16_bits=True))
t_lab, (ids, tab) = timed(lambda: ops.label_table(mask))
n = len(ids)
t_min, cmin = timed(lambda: ops.channel_min(image))
ids_d = torch.from_numpy(ids.astype(np.int32)).to(dev)
bb_d = torch.from_numpy(tab[:, :4].astype(np.int32)).to(dev)
t_patch, (patches, avg) = timed(lambda: ops.extract_patches(image, mask, cmin, ids_d, bb_d, want_avg=True))
pa = torch.softmax(torch.randn((n, 12), device=dev), 1)
pb = torch.softmax(torch.randn((n, 6), device=dev), 1)
from multiplexed_image_annotator_amd.annotator import CLASS_NAMES
gid = {name: i for i, name in enumerate(ops.GLOBAL_NAMES)}
t_vote, _ = timed(lambda: ops.vote(pa, [gid[c] for c in CLASS_NAMES["immune_full"]], pb, [gid[c] for c in CLASS_NAMES["struct"]], [-1.0] * 18, 0.3))
hw = S * S
rows = [
    ("P1 normalise (bg sigma 20 + blur 0.3 + exact percentile)", t_norm, (2 + 4) * C * hw),
    ("P2 label table", t_lab, 4 * hw),
    ("P3 channel minimum", t_min, 4 * C * hw),
    (f"P4-P7 crop + soft mask + intensity, {n} cells x {C} ch", t_patch, n * C * 1600 * 4),
    ("V vote", t_vote, n * (18 * 4 + 5)),
]
print(f"{'stage':62s} {'ms':>9s} {'algorithmic MB':>15s} {'GB/s':>9s} {'% of 8 TB/s':>12s}")
for name, t, b in rows:
    print(f"{name:62s} {t * 1e3:9.2f} {b / 1e6:15.1f} {b / t / 1e9:9.1f} {100 * b / t / 8e12:12.2f}")
print(f"sum {1e3 * sum(r[1] for r in rows):.1f} ms per tile = {100 * sum(r[1] for r in rows) / 7.8:.1f} % of a 7.8 s pass")
