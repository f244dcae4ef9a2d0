"""ORACLE (test infrastructure).  scikit-image is a dependency of the reference that is neither vendored
under /root/reference nor installed here (pyproject.toml:32, unpinned).  The four functions the hot path
calls are restated on the scipy.ndimage primitives scikit-image itself delegates to:

* ``skimage.morphology.disk`` / ``dilation``   (call site cell_type_annotation/utils.py:260)
* ``skimage.filters.gaussian``                 (call site cell_type_annotation/utils.py:265)
* ``skimage.transform.resize``                 (call site cell_type_annotation/preprocess.py:106)

These semantics are recalled from the published scikit-image sources (>= 0.19), not verifiable in this
container: the parts of the goldens that pass through them are "parity unpinned" for the third-party
arithmetic and pinned only for the reference's own control flow.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi


def disk(radius: int, dtype=np.uint8) -> np.ndarray:
    """Euclidean disk footprint {(dy,dx): dy^2+dx^2 <= radius^2}."""
    r = int(radius)
    yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
    return (yy * yy + xx * xx <= r * r).astype(dtype)


def dilation(image: np.ndarray, footprint: np.ndarray) -> np.ndarray:
    """Binary dilation of a bool image.  skimage runs ``ndi.grey_dilation`` with the mirrored footprint;
    for the symmetric convex disks used here border handling adds no pixels (a mirrored outside pixel is
    always farther from the probe than its in-image source), so a zero-border binary dilation is identical."""
    return ndi.binary_dilation(np.asarray(image, dtype=bool), structure=footprint.astype(bool))


def gaussian(image: np.ndarray, sigma: float) -> np.ndarray:
    """``filters.gaussian(image, sigma)`` defaults: bool/uint -> float64 in [0,1], ``mode='nearest'``,
    ``truncate=4.0``; separable fp64 filter, axis 0 then axis 1."""
    img = np.asarray(image)
    if img.dtype == bool:
        img = img.astype(np.float64)
    elif img.dtype.kind in "ui":
        img = img.astype(np.float64) / np.iinfo(img.dtype).max
    elif img.dtype == np.float16:
        img = img.astype(np.float32)
    return ndi.gaussian_filter(img, sigma, mode="nearest", truncate=4.0)


def resize(image: np.ndarray, output_shape, order: int = 0, anti_aliasing: bool = True, preserve_range: bool = True) -> np.ndarray:
    """``transform.resize(..., order=0, anti_aliasing=True, preserve_range=True)`` with the default
    ``mode='reflect'`` (numpy.pad naming, i.e. ndimage ``'mirror'``): optional Gaussian pre-filter with
    sigma = max(0, (factor-1)/2) per axis, then ``ndi.zoom(order=0, grid_mode=True)``.
    With factor 1 on every axis both steps are the identity (the reference's default cell_size=30)."""
    img = np.asarray(image)
    out_shape = tuple(int(s) for s in output_shape)
    if img.shape == out_shape:
        return img.astype(np.float64, copy=True) if img.dtype.kind != "f" else img.copy()
    factors = np.divide(img.shape, out_shape)
    filtered = img
    if anti_aliasing:
        sig = np.maximum(0, (factors - 1) / 2)
        if np.any(sig > 0):
            filtered = ndi.gaussian_filter(img, sig, cval=0, mode="mirror")
    zoom = [1.0 / f for f in factors]
    out = ndi.zoom(filtered, zoom, order=order, mode="mirror", cval=0, grid_mode=True)
    # resize(clip=True) default: _clip_warp_output clips to the range of the (unfiltered) input
    np.clip(out, np.min(img), np.max(img), out=out)
    return out
