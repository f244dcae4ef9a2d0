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
    when it is installed; otherwise the baseline-TIFF reader below (``read_tiff``: the layouts multiplexed-imaging exports use and PIL
    cannot read -- planar multi-sample, tiled, deflate + predictor, BigTIFF, OME-TIFF page stacks with pyramid levels), and PIL page by
    page for what that reader declines (LZW / JPEG compression).  ``.npy`` is an extension of this package (device-ready arrays)."""
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
        try:
            return read_tiff(p)
        except TiffUnsupported:
            pass
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


class TiffUnsupported(ValueError):
    """a TIFF feature ``read_tiff`` does not implement, or a file it cannot make sense of (the caller falls back to PIL)"""


class TiffStackError(ValueError):
    """the file is a Z / T stack: not the (C, H, W) image the hot path takes (tifffile would return a 4-D / 5-D array, which
    ``as_channel_planes`` rejects) -- never handed to a fallback reader that would flatten it into channels"""


_TIFF_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "B", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d", 13: "I", 16: "Q", 17: "q", 18: "Q"}


def _stack_dims(description: str, path: str):
    """(expected pages or None, images stored behind ONE IFD or None) from the first page's ImageDescription: OME-XML ``SizeC / SizeZ /
    SizeT`` and ImageJ ``images= / channels= / slices= / frames=``.  A description that announces Z or T planes raises TiffStackError."""
    import re
    if not description:
        return None, None
    if "<OME" in description and "<Pixels" in description:
        px = description[description.index("<Pixels"):]
        px = px[:px.index(">") + 1] if ">" in px else px
        dims = {}
        for key in ("SizeC", "SizeZ", "SizeT"):
            m = re.search(key + r'\s*=\s*"(\d+)"', px)
            dims[key] = int(m.group(1)) if m else 1
        if dims["SizeZ"] != 1 or dims["SizeT"] != 1:
            raise TiffStackError(f"{path}: OME stack with SizeZ = {dims['SizeZ']}, SizeT = {dims['SizeT']}: expected one (C, H, W) image")
        return dims["SizeC"], None
    if description.startswith("ImageJ="):
        kv = dict(line.split("=", 1) for line in description.splitlines() if "=" in line)

        def num(key, default):
            try:
                return int(kv.get(key, default))
            except ValueError:
                return default
        images, slices, frames = num("images", 1), num("slices", 1), num("frames", 1)
        if slices != 1 or frames != 1:
            raise TiffStackError(f"{path}: ImageJ hyperstack with slices = {slices}, frames = {frames}: expected one (C, H, W) image")
        return images, images
    return None, None


def read_tiff(path: str) -> np.ndarray:
    """Baseline TIFF 6.0 / BigTIFF reader in numpy + zlib, returning what ``tifffile.imread`` returns for the first series:

    * little- or big-endian, classic (II*\0 / MM\0*) or BigTIFF (version 43);
    * every top-level page that is NOT a reduced-resolution image (NewSubfileType bit 0: pyramid levels of OME / QPTIFF exports are
      skipped, SubIFDs are never followed) and has the first page's shape and dtype, in file order: (pages, H, W);
    * strips or tiles; compression none (1) or deflate (8 / 32946), optional horizontal differencing (Predictor 2);
    * 8 / 16 / 32-bit unsigned or signed integers and 32 / 64-bit floats; SamplesPerPixel S > 1 either planar
      (PlanarConfiguration 2 -> (S, H, W), the channel-first layout the hot path wants) or chunky ((H, W, S), as tifffile does);
    * an OME-XML ImageDescription must say SizeZ = SizeT = 1 and its SizeC must be the number of planes found; an ImageJ description
      must say slices = frames = 1, and its ``images=N`` planes may sit behind ONE IFD (contiguous, uncompressed: how ImageJ writes
      large stacks).  Z / T stacks raise ``TiffStackError`` (a plain ValueError for the caller: they are not (C, H, W) images).
    Anything else (LZW, JPEG, palette, sub-byte samples) and any file whose tags or strips cannot be decoded (truncated, missing
    StripOffsets, corrupt deflate stream) raises ``TiffUnsupported``.
    The file is memory-mapped and every page is decoded straight into the preallocated result: peak host memory is the image plus one
    strip or tile."""
    import struct
    import zlib
    try:
        return _read_tiff(path)
    except (TiffUnsupported, TiffStackError):
        raise
    except (struct.error, zlib.error, KeyError, IndexError, ValueError, OverflowError, TypeError) as e:
        raise TiffUnsupported(f"{path}: malformed or truncated TIFF ({type(e).__name__}: {e})") from e


def _read_tiff(path: str) -> np.ndarray:
    import struct
    import zlib
    if os.path.getsize(path) < 8:
        raise TiffUnsupported(f"{path}: not a TIFF file")
    data = np.memmap(path, dtype=np.uint8, mode="r")
    size = data.shape[0]

    def raw_bytes(o, n):
        if o < 0 or n < 0 or o + n > size:
            raise TiffUnsupported(f"{path}: bytes {o} .. {o + n} lie beyond the end of the file ({size})")
        return data[o:o + n].tobytes()

    head = raw_bytes(0, 8)
    if head[:2] not in (b"II", b"MM"):
        raise TiffUnsupported(f"{path}: not a TIFF file")
    bo = "<" if head[:2] == b"II" else ">"
    version = struct.unpack(bo + "H", head[2:4])[0]
    if version == 42:
        big, off = False, struct.unpack(bo + "I", head[4:8])[0]
    elif version == 43:
        big, off = True, struct.unpack(bo + "Q", raw_bytes(8, 8))[0]
    else:
        raise TiffUnsupported(f"{path}: unknown TIFF version {version}")

    def read_ifd(o):
        n = struct.unpack(bo + ("Q" if big else "H"), raw_bytes(o, 8 if big else 2))[0]
        o += 8 if big else 2
        esz, vsz = (20, 8) if big else (12, 4)
        table = raw_bytes(o, n * esz + vsz)
        tags = {}
        for i in range(n):
            e = table[i * esz:(i + 1) * esz]
            tag, typ = struct.unpack(bo + "HH", e[:4])
            cnt = struct.unpack(bo + ("Q" if big else "I"), e[4:4 + vsz])[0]
            fmt = _TIFF_TYPES.get(typ)
            if fmt is None:
                continue
            per = struct.calcsize("=" + fmt)
            nbytes = per * cnt
            if nbytes <= vsz:
                raw = e[4 + vsz:4 + vsz + nbytes]
            else:
                vo = struct.unpack(bo + ("Q" if big else "I"), e[4 + vsz:4 + 2 * vsz])[0]
                raw = raw_bytes(vo, nbytes)
            if typ == 2:
                tags[tag] = raw.rstrip(b"\0").decode("latin-1")
            else:
                tags[tag] = struct.unpack(bo + fmt * cnt, raw)
        nxt = struct.unpack(bo + ("Q" if big else "I"), table[n * esz:n * esz + vsz])[0]
        return tags, nxt

    def one(tags, tag, default=None):
        v = tags.get(tag)
        return default if v is None else v[0]

    def page_layout(tags):
        """everything needed to decode a page, from its tags alone"""
        w, h = one(tags, 256), one(tags, 257)
        spp = one(tags, 277, 1)
        bits = tags.get(258, (1,))
        fmt = one(tags, 339, 1)
        comp = one(tags, 259, 1)
        planar = one(tags, 284, 1)
        pred = one(tags, 317, 1)
        if not w or not h or len(set(bits)) != 1 or bits[0] not in (8, 16, 32, 64):
            raise TiffUnsupported(f"{path}: unsupported sample layout {bits}")
        if comp not in (1, 8, 32946):
            raise TiffUnsupported(f"{path}: compression {comp} is not implemented")
        if one(tags, 262, 1) == 3:
            raise TiffUnsupported(f"{path}: palette images are not implemented")
        kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
        if kind is None or (kind == "f" and bits[0] < 32) or pred not in (1, 2) or (pred == 2 and kind == "f"):
            raise TiffUnsupported(f"{path}: sample format {fmt} / predictor {pred}")
        dt = np.dtype(bo + kind + str(bits[0] // 8))
        tiled = 322 in tags
        if tiled:
            tw, th = one(tags, 322), one(tags, 323)
            offs, cnts = tags.get(324), tags.get(325)
        else:
            tw, th = w, min(one(tags, 278, h), h)
            offs, cnts = tags.get(273), tags.get(279)
        if offs is None or cnts is None or not tw or not th:
            raise TiffUnsupported(f"{path}: a page without strip / tile offsets and byte counts")
        across, down = (w + tw - 1) // tw, (h + th - 1) // th
        planes = spp if planar == 2 else 1
        per_chunk = spp if planar == 1 else 1
        if len(offs) != across * down * planes or len(cnts) != len(offs):
            raise TiffUnsupported(f"{path}: {len(offs)} strips / tiles, expected {across * down * planes}")
        if planar == 2 and spp > 1:
            shape = (spp, h, w)
        elif spp > 1:
            shape = (h, w, spp)
        else:
            shape = (h, w)
        return dict(w=w, h=h, spp=spp, comp=comp, planar=planar, pred=pred, dt=dt, tiled=tiled, tw=tw, th=th, offs=offs, cnts=cnts,
                    across=across, down=down, planes=planes, per_chunk=per_chunk, shape=shape)

    def decode_into(L, dst):
        """dst: the page's slot of the result, shape L['shape'], native byte order"""
        dt, nat = L["dt"], L["dt"].newbyteorder("=")
        h, w, tw, th = L["h"], L["w"], L["tw"], L["th"]
        view = dst.reshape((L["planes"], h, w, L["per_chunk"])) if L["planar"] == 2 or L["spp"] == 1 else dst.reshape((1, h, w, L["spp"]))
        for idx, (o, c) in enumerate(zip(L["offs"], L["cnts"])):
            pl, rem = divmod(idx, L["across"] * L["down"])
            ty, tx = divmod(rem, L["across"])
            rows = th if L["tiled"] else min(th, h - ty * th)
            want = rows * tw * L["per_chunk"]
            if L["comp"] != 1:
                buf = zlib.decompress(raw_bytes(o, c))
                if len(buf) < want * dt.itemsize:
                    raise TiffUnsupported(f"{path}: a strip / tile decompresses to {len(buf)} bytes, expected {want * dt.itemsize}")
                chunk = np.frombuffer(buf, dtype=dt, count=want)
            else:
                if c < want * dt.itemsize or o + want * dt.itemsize > size:
                    raise TiffUnsupported(f"{path}: a strip / tile of {c} bytes at {o}, expected {want * dt.itemsize} inside the file")
                chunk = data[o:o + want * dt.itemsize].view(dt)
            chunk = chunk.reshape(rows, tw, L["per_chunk"])
            if L["pred"] == 2:
                chunk = np.cumsum(chunk.astype(nat), axis=1, dtype=nat)
            y0, x0 = ty * th, tx * tw
            y1, x1 = min(y0 + rows, h), min(x0 + tw, w)
            view[pl, y0:y1, x0:x1] = chunk[:y1 - y0, :x1 - x0]

    layouts, seen, description = [], set(), None
    expect, behind_one_ifd = None, None
    while off and off not in seen and off + 2 <= size:
        seen.add(off)
        tags, off = read_ifd(off)
        if description is None:
            description = tags.get(270, "") if isinstance(tags.get(270, ""), str) else ""
            # BEFORE any page layout is looked at: a Z / T stack in a compression this reader does not decode (LZW, JPEG) must be refused
            # as a stack (TiffStackError: no fallback), not as "unsupported" -- read_image() would hand that to PIL, which stacks the Z / T
            # planes as channels
            expect, behind_one_ifd = _stack_dims(description, path)
        if one(tags, 254, 0) & 1:                    # reduced-resolution page of a pyramid
            continue
        layouts.append(page_layout(tags))
    if not layouts:
        raise TiffUnsupported(f"{path}: no image pages")
    first = layouts[0]
    same = [L for L in layouts if L["shape"] == first["shape"] and L["dt"] == first["dt"]]
    nat = first["dt"].newbyteorder("=")
    if behind_one_ifd and behind_one_ifd > 1 and len(same) == 1:
        # ImageJ's contiguous stack: one IFD, `images` planes of the first page's size following its first strip
        n = behind_one_ifd
        plane = int(np.prod(first["shape"])) * first["dt"].itemsize
        if first["comp"] != 1 or first["tiled"] or first["spp"] != 1:
            raise TiffUnsupported(f"{path}: ImageJ stack of {n} images behind one IFD, not plain uncompressed strips")
        o = first["offs"][0]
        if o + n * plane > size:
            raise TiffUnsupported(f"{path}: ImageJ stack of {n} images does not fit the file")
        out = np.empty((n,) + first["shape"], dtype=nat)
        out[...] = data[o:o + n * plane].view(first["dt"]).reshape(out.shape)
        return out
    if expect is not None:
        planes_found = len(same) * (first["spp"] if first["spp"] > 1 else 1)
        if planes_found > expect and first["spp"] == 1:
            # more same-shape pages than the description's SizeC / images: the first series is the first `expect` of them (what tifffile's
            # imread returns for such a file); the extra pages belong to something else (a second series, a thumbnail without the flag)
            same = same[:expect]
        elif planes_found != expect:
            # FEWER planes than announced (or a count that pixel-interleaved pages cannot make up): e.g. a multi-file OME dataset whose
            # SizeC counts planes held in companion files, which this reader does not follow.  No fallback either: another decoder reading
            # page by page would return a partial image as if it were complete.  Stricter than the reference's tifffile path, which follows
            # the companions (deviations).
            raise TiffStackError(f"{path}: the description announces {expect} channel planes, the file holds {planes_found} "
                                 f"({len(same)} page(s) of shape {first['shape']}, {first['dt']}; {len(layouts) - len(same)} page(s) of another shape): "
                                 "planes kept in companion files of a multi-file OME dataset are not read -- export one (C, H, W) file")
    if len(same) == 1:
        out = np.zeros(first["shape"], dtype=nat)
        decode_into(first, out)
        return out
    out = np.zeros((len(same),) + first["shape"], dtype=nat)
    for k, L in enumerate(same):
        decode_into(L, out[k])
    return out


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
        self.cell_ids_dev: List[torch.Tensor] = []          # per image: ascending cell ids int32 [n] / boxes int32 [n, 4] on the device
        self.cell_bbox_dev: List[torch.Tensor] = []
        # multi-rank runs (set by Annotator.preprocess): image_ids[j] = row of the batch CSV that local position j holds (tile-per-rank mode
        # keeps only this rank's images; every per-image list above is indexed by local position); norm_shard = (rank, world_size) when the
        # whole-image normalisation is split by channel with one all-gather of the planes (cell-sharded mode, RIBCA_NORM_SHARD=1)
        self.image_ids: List[int] = []
        self.norm_shard: Optional[Tuple[int, int]] = None
        self._log("\n")
        self._log("Starting image processing...")

    def _log(self, msg):
        if self.logger is not None:
            self.logger.log(msg)

    # ---- stage kernels ---------------------------------------------------------------------------------------------
    def _normalize(self, img, blur=0, amax=100) -> torch.Tensor:
        if self.norm_shard is None:
            return ops.normalize_image(img, blur=blur, amax=amax)
        # every channel is normalised on its own (background filter, percentile, scale: preprocess.py:216-238), so a rank can do
        # ceil(C / world) of them and the planes are exchanged once -- bit-identical to the replicated form
        from . import dist
        rank, ws = self.norm_shard
        c = img.shape[0]
        lo, hi = dist.shard_bounds(c, rank, ws)
        if hi > lo:
            local = ops.normalize_image(img[lo:hi], blur=blur, amax=amax)
        else:
            local = torch.empty((0,) + tuple(img.shape[1:]), dtype=torch.float32, device=_lib.require_gpu())
        return dist.all_gather_planes(local, c)

    def _cell_pos_dict(self, mask, n_jobs=0) -> LazyCellPositions:
        mask_np = np.asarray(mask).astype(np.int32)
        ids, table = ops.label_table(torch.from_numpy(mask_np).to(_lib.require_gpu()))
        return LazyCellPositions(mask_np, ids, table)

    def _move_image_range(self, image: torch.Tensor):
        """Per-channel minimum (C,) on the device; the shifted image is never materialised (the patch kernel subtracts it)."""
        return ops.channel_min(image), None

    def crop_cells(self, image_idx: int, lo: int, hi: int, want_avg: bool = False, out: Optional[torch.Tensor] = None):
        """Soft-masked full-channel patches of cells [lo, hi) (cell order = ascending id) of one image (into ``out`` when given)."""
        # the ids and boxes never left the device (transform keeps the label table's device copy beside the host one the CSV needs)
        ids = self.cell_ids_dev[image_idx][lo:hi].contiguous()
        bbox = self.cell_bbox_dev[image_idx][lo:hi].contiguous()
        return ops.extract_patches(self.images_dev[image_idx], self.masks_dev[image_idx], self.chan_min[image_idx], ids, bbox,
                                   want_avg=want_avg, out=out, patch_size=self.patch_size)

    # ---- reference entry point -------------------------------------------------------------------------------------
    def transform(self, shard_fn=None, gather_fn=None, keep_patches: bool = True, chunk: int = 16384, image_filter=None):
        """preprocess.py:241-290.  ``shard_fn(n) -> (lo, hi)`` picks this rank's cells (default: all);
        ``gather_fn(local (n_local, C) fp64 tensor, n) -> (n, C)`` reassembles per-cell rows across ranks;
        ``image_filter(i) -> bool`` keeps only some rows of the batch CSV (tile-per-rank mode: ``image_ids`` records which)."""
        dev = _lib.require_gpu()
        for i, (image_path, mask_path) in enumerate(zip(self.image_paths, self.mask_paths)):
            if image_filter is not None and not image_filter(i):
                continue
            self.image_ids.append(i)
            j = len(self.image_ids) - 1                  # LOCAL position: every per-image list below is indexed by it
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
            ids, table, ids_d, bbox_d = ops.label_table(mask_d, with_device=True)
            self.cell_ids_dev.append(ids_d)
            self.cell_bbox_dev.append(bbox_d)
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
                _, a = self.crop_cells(j, c0, c1, want_avg=True, out=kept[c0 - lo:c1 - lo] if keep_patches else None)
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
