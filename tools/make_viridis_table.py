#!/usr/bin/env python3
"""Regenerates multiplexed-image-annotator_amd/viridis_u8.npy: matplotlib's 256-entry viridis table, components truncated to
int(c * 255) exactly as the reference's utils.number_to_rgb does (cell_type_annotation/utils.py:16-28)."""
import os

import matplotlib
import numpy as np

matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402

cmap = plt.get_cmap("viridis")
lut = np.array([[int(c * 255) for c in cmap(i)[:3]] for i in range(256)], np.uint8)
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multiplexed-image-annotator_amd", "viridis_u8.npy"), lut)
print(lut.shape, lut[0], lut[-1])
