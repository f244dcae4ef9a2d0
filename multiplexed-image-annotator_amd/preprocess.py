"""``ImageProcessor``: GPU drop-in for the reference's pre-processing class (cell_type_annotation/preprocess.py:25-290).

Same constructor arguments and the attributes other code reads afterwards (``cell_pos_dict``, ``intensity_full``,
``masks``, ``_n_images``, ``save_path``).  What changes is where the work runs and where patches live:

* ``_normalize``       -> HIP kernels (ops.normalize_image), result stays on the device;
* ``_cell_pos_dict``   -> one GPU pass builds the per-label table (bbox, centroid sums, counts); the dict of pixel lists
  the reference builds eagerly is exposed as a lazy mapping (only post-analysis code outside the hot path reads lists);
* ``_img2patches``     -> one workgroup per cell crops / soft-masks all image channels once; every panel's tensor is a
  channel view of that (the reference recomputes the soft mask per panel and round-trips ``.pt`` files through
  ``main_dir/tmp``; here patches never leave HBM, the tmp directory is only created and cleared for compatibility).

Cells can be sharded across ranks (``shard=(lo, hi)`` in cell order): each rank crops only its own cells.
"""
from __future__ import annotations

import os
from collections.abc import Mapping
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops


def read_image(path: str) -> np.ndarray:
    """Image / mask reader standing in for ``skimage.io.imread`` (reference preprocess.py:244-246), which hands ``.tif`` files to
    tifffile: a multi-page TIFF comes back as (pages, H, W) in page order, a PNG as (H, W) or (H, W, samples).  ``tifffile`` is used
    when it is installed (it also reads the planar / OME / BigTIFF layouts PIL cannot); otherwise PIL page by page.  ``.npy`` is an
    extension of this package (device-ready arrays)."""
    p = str(path)
    low = p.lower()
    if low.endswith(".npy"):
        return np.load(p)
    if low.endswith((".tif", ".tiff")):
        try:
            import tifffile
        except ImportError:
            tifffile = None
        if tifffile is not None:
            return np.asarray(tifffile.imread(p))
    from PIL import Image
    im = Image.open(p)
    frames = getattr(im, "n_frames", 1)
    if frames > 1:
        planes = []
        for i in range(frames):
            im.seek(i)
            planes.append(np.array(im))
        if len({pl.shape for pl in planes}) != 1:
            raise ValueError(f"{p}: pages of different shapes cannot be stacked into (C, H, W)")
        return np.stack(planes, axis=0)
    return np.array(im)


def as_channel_planes(image: np.ndarray, path: str = "") -> np.ndarray:
    """The hot path indexes the image as (C, H, W) (reference preprocess.py:214-239 loops ``for i in range(img.shape[0])``): a single
    2-D page becomes one plane; anything that is not 3-D afterwards is rejected here instead of failing inside a kernel."""
    if image.ndim == 2:
        image = image[None]
    if image.ndim != 3:
        raise ValueError(f"{path}: expected an image of shape (C, H, W), got {image.shape}")
    return image


class LazyCellPositions(Mapping):
    """label -> (row list, col list) in raster order, keys ascending -- what ``_cell_pos_dict`` returns in the reference
    (preprocess.py:159-181) -- materialised per key on demand from the host copy of the mask."""

    def __init__(self, mask: np.ndarray, ids: np.ndarray, table: np.ndarray):
        self._mask = mask
        self.ids = ids
        self.table = table
        self._index: Optional[Tuple[np.ndarray, np.ndarray]] = None
        self._where = {int(k): i for i, k in enumerate(ids.tolist())}

    def __len__(self):
        return len(self.ids)

    def __iter__(self):
        return iter(self.ids.tolist())

    def __contains__(self, key):
        return int(key) in self._where

    def _build(self):
        flat = self._mask.ravel()
        idx = np.flatnonzero(flat)
        order = np.argsort(flat[idx], kind="stable")
        starts = np.concatenate(([0], np.cumsum(self.table[:, 6])))
        self._index = (idx[order], starts)

    def __getitem__(self, key):
        j = self._where[int(key)]
        if self._index is None:
            self._build()
        idx, starts = self._index
        px = idx[starts[j]:starts[j + 1]]
        w = self._mask.shape[1]
        return (px // w).tolist(), (px % w).tolist()


class ImageProcessor(object):
    def __init__(self, csv_path, parser, main_path, device, batch_id='', infer=True, normalization=True, blur=0, amax=100, cell_size=30,
                 logger=None, n_jobs=0) -> None:
        import pandas as pd
        table = pd.read_csv(csv_path)
        self.image_paths = table['image_path']
        self.mask_paths = table['mask_path']
        assert len(self.image_paths) == len(self.mask_paths)
        self.logger = logger
        self._n_images = len(self.image_paths)
        self._log("Number of images: {}.".format(self._n_images))
        self.main_dir = main_path
        self.save_path = os.path.join(self.main_dir, "tmp")
        self.batch_id = batch_id
        self.normalization = normalization
        self.blur = blur
        self.amax = amax
        self.parser = parser
        self.cell_pos_dict: List[LazyCellPositions] = []
        self.intensity_full: List[Optional[np.ndarray]] = []
        os.makedirs(self.save_path, exist_ok=True)
        for name in os.listdir(self.save_path):          # the reference clears its patch spill directory here
            full = os.path.join(self.save_path, name)
            if os.path.isfile(full):
                os.remove(full)
        self.infer = infer
        self.masks: List[np.ndarray] = []
        self.device = device
        self.scale = cell_size / 30.0
        self.patch_size = int(40 * self.scale)           # preprocess.py:78
        if not 4 <= self.patch_size <= 90:
            raise NotImplementedError("cell_size must give a crop window of 4..90 pixels (cell_size 3..67): one workgroup holds the window")
        self.n_jobs = n_jobs
        # device-resident state of the hot path
        self.images_dev: List[torch.Tensor] = []
        self.masks_dev: List[torch.Tensor] = []
        self.chan_min: List[torch.Tensor] = []
        self.cell_ids: List[np.ndarray] = []
        self.cell_tables: List[np.ndarray] = []
        self.shards: List[Tuple[int, int]] = []
        self.patches: List[Optional[torch.Tensor]] = []     # (n_local, C_img, 40, 40) fp32 per image, this rank's cells
        self._log("\n")
        self._log("Starting image processing...")

    def _log(self, msg):
        if self.logger is not None:
            self.logger.log(msg)

    # ---- stage kernels ---------------------------------------------------------------------------------------------
    def _normalize(self, img, blur=0, amax=100) -> torch.Tensor:
        return ops.normalize_image(img, blur=blur, amax=amax)

    def _cell_pos_dict(self, mask, n_jobs=0) -> LazyCellPositions:
        mask_np = np.asarray(mask).astype(np.int32)
        ids, table = ops.label_table(torch.from_numpy(mask_np).to(_lib.require_gpu()))
        return LazyCellPositions(mask_np, ids, table)

    def _move_image_range(self, image: torch.Tensor):
        """Per-channel minimum (C,) on the device; the shifted image is never materialised (the patch kernel subtracts it)."""
        return ops.channel_min(image), None

    def crop_cells(self, image_idx: int, lo: int, hi: int, want_avg: bool = False, out: Optional[torch.Tensor] = None):
        """Soft-masked full-channel patches of cells [lo, hi) (cell order = ascending id) of one image (into ``out`` when given)."""
        dev = self.images_dev[image_idx].device
        ids = torch.from_numpy(self.cell_ids[image_idx][lo:hi].astype(np.int32)).to(dev)
        bbox = torch.from_numpy(self.cell_tables[image_idx][lo:hi, :4].astype(np.int32)).to(dev)
        return ops.extract_patches(self.images_dev[image_idx], self.masks_dev[image_idx], self.chan_min[image_idx], ids, bbox,
                                   want_avg=want_avg, out=out, patch_size=self.patch_size)

    # ---- reference entry point -------------------------------------------------------------------------------------
    def transform(self, shard_fn=None, gather_fn=None, keep_patches: bool = True, chunk: int = 16384):
        """preprocess.py:241-290.  ``shard_fn(n) -> (lo, hi)`` picks this rank's cells (default: all);
        ``gather_fn(local (n_local, C) fp64 tensor, n) -> (n, C)`` reassembles per-cell rows across ranks."""
        dev = _lib.require_gpu()
        for i, (image_path, mask_path) in enumerate(zip(self.image_paths, self.mask_paths)):
            image = as_channel_planes(read_image(image_path), image_path)
            mask = read_image(mask_path)
            if mask.ndim == 3:
                mask = mask[:, :, 0]                      # the reference assumes the first channel holds the labels
            mask = mask.astype(np.int32)
            if self.normalization:
                img_d = self._normalize(image, blur=self.blur, amax=self.amax)
            else:
                img_d = torch.from_numpy(np.ascontiguousarray(image).astype(np.float32)).to(dev)
            self.masks.append(mask)
            mask_d = torch.from_numpy(mask).to(dev)
            ids, table = ops.label_table(mask_d)
            self.cell_pos_dict.append(LazyCellPositions(mask, ids, table))
            self.images_dev.append(img_d)
            self.masks_dev.append(mask_d)
            self.chan_min.append(ops.channel_min(img_d))
            self.cell_ids.append(ids)
            self.cell_tables.append(table)
            n = len(ids)
            lo, hi = shard_fn(n) if shard_fn is not None else (0, n)
            self.shards.append((lo, hi))
            any_panel = any(self.parser.indices.get(p) is not None for p in self.parser.panels)
            if not any_panel:
                self.patches.append(None)
                continue
            # intensity table (preprocess.py:138-149) comes with the crop; it is panel independent (all image channels).
            # Patches are cropped straight into ONE preallocated tensor (9.6 GB at 100 k cells x 15 channels): no per-chunk
            # pieces to concatenate, so the peak footprint is the tensor itself.
            c_img = img_d.shape[0]
            kept = torch.empty((hi - lo, c_img, 40, 40), dtype=torch.float32, device=dev) if keep_patches else None
            avg = torch.empty((hi - lo, c_img), dtype=torch.float64, device=dev)
            for c0 in range(lo, hi, chunk):
                c1 = min(c0 + chunk, hi)
                _, a = self.crop_cells(i, c0, c1, want_avg=True, out=kept[c0 - lo:c1 - lo] if keep_patches else None)
                avg[c0 - lo:c1 - lo] = a
            if gather_fn is not None:
                avg = gather_fn(avg, n)
            self.intensity_full.append((avg.cpu().numpy() + 1) / 2 if n else None)
            self.patches.append(kept)

    def panel_patches(self, image_idx: int, lo: Optional[int] = None, hi: Optional[int] = None) -> torch.Tensor:
        """This rank's full-channel patches of one image (cached by transform, or cropped now for a sub-range)."""
        slo, shi = self.shards[image_idx]
        if lo is None:
            lo, hi = slo, shi
        cached = self.patches[image_idx]
        if cached is not None:
            return cached[lo - slo:hi - slo]
        return self.crop_cells(image_idx, lo, hi)[0]
