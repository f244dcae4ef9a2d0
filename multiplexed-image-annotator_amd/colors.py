"""Palette helpers of the label-painting step (host side, tiny tables): what the reference's ``utils.get_colors``
(cell_type_annotation/utils.py:33-107) and ``utils.number_to_rgb`` (utils.py:16-28) return, as arrays."""
from __future__ import annotations

import colorsys
import os
from typing import List, Tuple

import numpy as np

# utils.py:47-67: the first 18 palette entries are fixed; "Others" (always the last cell type) is silver
_FIXED = np.array([[255, 0, 0], [0, 0, 255], [0, 128, 0], [255, 255, 0], [255, 0, 255], [0, 255, 255], [255, 165, 0], [128, 0, 128],
                   [0, 128, 128], [128, 0, 0], [0, 0, 128], [128, 128, 0], [255, 192, 203], [165, 42, 42], [0, 255, 0], [135, 206, 235],
                   [75, 0, 130], [255, 215, 0], [192, 192, 192]], dtype=np.uint8)
SILVER = (192, 192, 192)
_GOLDEN = 0.618033988749895
_LEVELS = (0.7, 0.8, 0.9, 1.0)
_VIRIDIS = None


def get_colors(n: int) -> List[Tuple[int, int, int]]:
    """n colours: n-1 distinct ones followed by silver.  Beyond the fixed table hues advance by the golden ratio from 0.1 and
    saturation = value cycle through 0.7 .. 1.0 with the running count, components truncated to int(c * 255)."""
    want = n - 1
    out = [tuple(int(c) for c in row) for row in _FIXED[:max(0, min(want, len(_FIXED)))]]
    hue = 0.1
    while len(out) < want:
        hue = (hue + _GOLDEN) % 1.0
        level = _LEVELS[len(out) % len(_LEVELS)]
        out.append(tuple(int(c * 255) for c in colorsys.hsv_to_rgb(hue, level, level)))
    out.append(SILVER)
    return out


def viridis_table() -> np.ndarray:
    """(256, 3) uint8: matplotlib's viridis look-up table with each component truncated to int(c * 255) (utils.py:24-26).
    Shipped as data (viridis_u8.npy, generated from matplotlib by tools/make_viridis_table.py)."""
    global _VIRIDIS
    if _VIRIDIS is None:
        _VIRIDIS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "viridis_u8.npy"))
    return _VIRIDIS


def confidence_colors(conf: np.ndarray) -> np.ndarray:
    """(n, 3) uint8 colour per cell: viridis at index int(conf * 256) (256 -> 255) for positive confidences, silver for the
    thresholded ones (-1), as model.py:831."""
    conf = np.asarray(conf, dtype=np.float32)
    idx = np.minimum((np.clip(conf, 0, 1) * np.float32(256)).astype(np.int64), 255)
    rgb = viridis_table()[idx]
    rgb[~(conf > 0)] = SILVER
    return rgb
