"""ORACLE (test infrastructure).  CPU restatement of ``spatial_methods.neighborhood_analysis``
(cell_type_annotation/spatial_methods.py:13-130, reached from ``Annotator.neighborhood_analysis``, model.py:798-800): for every
cell the n_neighbors nearest centroids (itself first), a cell-type x cell-type count of (cell, neighbour) pairs over the other
n_neighbors - 1, optional row normalisation, and the CSV text.  The heat-map PNG (seaborn) is not part of the data.
Pinned by tests/golden/neighborhood.json, produced by the reference's own function with scikit-learn's ball tree."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def centroids(table: np.ndarray):
    """x = mean(Column), y = mean(Row) per cell from the label table (sum_r, sum_c, count in columns 4..6)."""
    return table[:, 5].astype(np.float64) / table[:, 6].astype(np.float64), table[:, 4].astype(np.float64) / table[:, 6].astype(np.float64)


def cooccurrence(x: np.ndarray, y: np.ndarray, types: np.ndarray, n_types: int, n_neighbors: int) -> np.ndarray:
    """Brute-force k nearest neighbours in fp64 (squared distance dx*dx + dy*dy, ties by lower index), first hit dropped."""
    n = len(x)
    if n_neighbors > n:
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {n_neighbors}, n_samples_fit = {n}")
    out = np.zeros((n_types, n_types), np.float64)
    for j in range(n):
        dx, dy = x - x[j], y - y[j]
        d = dx * dx + dy * dy
        order = np.lexsort((np.arange(n), d))[:n_neighbors]
        for k in order[1:]:
            out[types[j], types[k]] += 1
    return out


def normalize_rows(m: np.ndarray) -> np.ndarray:
    m = m.copy()
    for i in range(len(m)):
        if m[i].sum() > 0:
            m[i] /= m[i].sum()
    return m


def csv_text(m: np.ndarray, cell_types: Sequence[str]) -> str:
    """spatial_methods.py:57-69: header row, then one row per type, every value ``%.3f`` followed by a comma."""
    lines = ["cell_type," + "".join(f"{c}," for c in cell_types) + "\n"]
    for i, c in enumerate(cell_types):
        lines.append(f"{c}," + "".join(f"{m[i][j]:.3f}," for j in range(len(cell_types))) + "\n")
    return "".join(lines)


TISSUE_NEIGHBOURHOODS = (10, 20, 30, 50, 75, 100, 150, 200)       # spatial_methods.py:155


def compositions(x: np.ndarray, y: np.ndarray, types: np.ndarray, sizes: Sequence[int] = TISSUE_NEIGHBOURHOODS) -> np.ndarray:
    """spatial_methods.py:157-176: per cell, for each neighbourhood size, the fraction of every cell type (0 .. max type) among
    its nearest other cells (201-NN query, the cell itself dropped), concatenated size-major.  Brute force, ties by lower index."""
    n = len(x)
    k = max(sizes) + 1
    if k > n:
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, n_samples_fit = {n}")
    n_types = int(np.max(types)) + 1
    out = np.zeros((n, len(sizes) * n_types), np.float64)
    for j in range(n):
        dx, dy = x - x[j], y - y[j]
        d = dx * dx + dy * dy
        order = np.lexsort((np.arange(n), d))[1:k]
        row = []
        for s in sizes:
            temp = np.bincount(types[order[:s]], minlength=n_types).astype(np.float64)
            temp /= np.sum(temp)
            row.extend(temp)
        out[j] = row
    return out
