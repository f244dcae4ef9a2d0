"""ORACLE (test infrastructure).  CPU restatement of the label-painting consumer of predict():
``Annotator.colorize`` (cell_type_annotation/model.py:806-858, without tissue regions), ``utils.get_colors``
(utils.py:33-107) and ``utils.number_to_rgb`` (utils.py:16-28).  Pinned by tests/golden/colorize.npz, produced by the
reference's own functions in the build container (tests/golden/make_golden.py::golden_colorize)."""
from __future__ import annotations

import colorsys
from typing import List, Sequence, Tuple

import numpy as np

#: utils.py:47-67 -- the fixed part of the palette
STANDARD = [(255, 0, 0), (0, 0, 255), (0, 128, 0), (255, 255, 0), (255, 0, 255), (0, 255, 255), (255, 165, 0), (128, 0, 128),
            (0, 128, 128), (128, 0, 0), (0, 0, 128), (128, 128, 0), (255, 192, 203), (165, 42, 42), (0, 255, 0), (135, 206, 235),
            (75, 0, 130), (255, 215, 0), (192, 192, 192)]
GRAY = (192, 192, 192)


def get_colors(n: int) -> List[Tuple[int, int, int]]:
    """utils.py:33-107: n-1 palette colours (fixed table, then golden-ratio hues with cycling saturation / value) + gray last."""
    n = n - 1
    if n <= len(STANDARD):
        return list(STANDARD[:n]) + [GRAY]
    colors = list(STANDARD)
    h = 0.1
    levels = [0.7, 0.8, 0.9, 1.0]
    while len(colors) < n:
        h = (h + 0.618033988749895) % 1.0
        s = levels[len(colors) % 4]
        v = levels[len(colors) % 4]
        r, g, b = colorsys.hsv_to_rgb(h, s, v)
        colors.append((int(r * 255), int(g * 255), int(b * 255)))
    colors.append(GRAY)
    return colors


def viridis_rgb(values: np.ndarray) -> np.ndarray:
    """utils.py:16-28 for an array: matplotlib's 256-entry viridis table at index int(v * 256) (256 -> 255), each component
    truncated to int(c * 255)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    cmap = plt.get_cmap("viridis")
    out = np.zeros((len(values), 3), np.int64)
    for i, v in enumerate(values):
        out[i] = [int(x * 255) for x in cmap(float(v))[:3]]
    return out


def colorize(mask: np.ndarray, ids: Sequence[int], labels: Sequence[str], conf: Sequence[float], cell_types: Sequence[str]):
    """model.py:806-858 with n_regions = 0: (H, W, 3) uint8 cell-type colours, (H, W, 3) uint8 confidence colours (gray where the
    confidence is not positive), (H, W) uint8 cell-type index + 1; background stays 0."""
    colors = get_colors(len(cell_types))
    types = list(cell_types)
    h, w = mask.shape
    type_rgb = np.zeros((h, w, 3), np.uint8)
    conf_rgb = np.zeros((h, w, 3), np.uint8)
    type_idx = np.zeros((h, w), np.uint8)
    pos = [c for c in conf if c > 0]
    vir = viridis_rgb(np.array(pos, np.float32)) if pos else np.zeros((0, 3), np.int64)
    k = 0
    for j, key in enumerate(ids):
        t = types.index(labels[j])
        sel = mask == key
        type_rgb[sel] = colors[t]
        if conf[j] > 0:
            conf_rgb[sel] = vir[k]
            k += 1
        else:
            conf_rgb[sel] = GRAY
        type_idx[sel] = t + 1
    return type_rgb, conf_rgb, type_idx
