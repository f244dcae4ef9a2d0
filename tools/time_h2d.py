#!/usr/bin/env python3
"""PCIe leg of the boundary: host uint16 image (15 x 4096 x 4096) + int32 mask -> device, and the
Annotator-level wall time (file read + H2D + preprocess + predict + CSV) next to the HBM-resident figure bench.py reports."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from multiplexed_image_annotator_amd import _lib

dev = _lib.require_gpu()
raw = np.zeros((15, 4096, 4096), np.uint16)
mask = np.zeros((4096, 4096), np.int32)
for pin in (False, True):
    a = torch.from_numpy(raw)
    b = torch.from_numpy(mask)
    if pin:
        a, b = a.pin_memory(), b.pin_memory()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        x = a.to(dev, non_blocking=True)
        y = b.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    gb = (raw.nbytes + mask.nbytes) / 1e9
    print(f"H2D {'pinned' if pin else 'pageable'}: {best * 1e3:.1f} ms for {gb:.3f} GB = {gb / best:.1f} GB/s")
