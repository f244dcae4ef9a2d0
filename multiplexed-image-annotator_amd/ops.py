"""Thin Python wrappers over the C ABI: one function per entry point of ``include/ribca_hip.h``.

Inputs and outputs are torch CUDA(HIP) tensors; nothing here computes on the CPU except tiny host-side
parameter preparation (Gaussian taps, index tables).  All calls enqueue on the current torch stream.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

PATCH = 40
OTHERS = 17
#: utils.py:143-146 key order (tie-break order of the vote); index = global class id, 17 = Others
VOTE_ORDER: List[str] = ["CD4 T cell", "CD8 T cell", "Dendritic cell", "B cell", "M1 macrophage cell", "M2 macrophage cell",
                         "Regulatory T cell", "Granulocyte cell", "Plasma cell", "Natural killer cell", "Mast cell",
                         "Stroma cell", "Smooth muscle", "Endothelial cell", "Epithelial cell", "Proliferating/tumor cell",
                         "Nerve cell"]
GLOBAL_NAMES: List[str] = VOTE_ORDER + ["Others"]

#: state-dict key order of the flat parameter blob (include/ribca_hip.h, ribca_vit_blob_len)
def blob_keys(depth: int) -> List[str]:
    keys = ["cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"]
    for i in range(depth):
        p = f"blocks.{i}."
        keys += [p + "norm1.weight", p + "norm1.bias", p + "attn.qkv.weight", p + "attn.qkv.bias", p + "attn.proj.weight",
                 p + "attn.proj.bias", p + "norm2.weight", p + "norm2.bias", p + "mlp.fc1.weight", p + "mlp.fc1.bias",
                 p + "mlp.fc2.weight", p + "mlp.fc2.bias"]
    return keys + ["norm.weight", "norm.bias", "head.weight", "head.bias"]


def gaussian_taps() -> np.ndarray:
    """27 fp64 weights for sigma = 1, 2, 3 (truncate 4.0), computed exactly as scipy.ndimage's
    ``_gaussian_kernel1d`` does (reference utils.py:265 -> skimage.filters.gaussian -> scipy): entry k of each
    segment is the normalised weight at distance k."""
    out = []
    for sigma in (1, 2, 3):
        sd = float(sigma)
        radius = int(4.0 * sd + 0.5)
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (sd * sd) * x ** 2)
        phi = phi / phi.sum()
        out.append(phi[radius:])
    taps = np.concatenate(out)
    assert taps.shape == (27,)
    return taps


_TAPS_DEV: Dict[int, torch.Tensor] = {}
_WS: Dict[int, torch.Tensor] = {}


def _taps(device) -> torch.Tensor:
    key = device.index or 0
    if key not in _TAPS_DEV:
        _TAPS_DEV[key] = torch.from_numpy(gaussian_taps()).to(device)
    return _TAPS_DEV[key]


def workspace(nbytes: int, device, slot: int = 0) -> torch.Tensor:
    """Grow-only scratch buffer per (device, slot) (caller-owned memory of the C ABI).  Work enqueued on different streams
    at the same time must use different slots."""
    key = (device.index or 0, slot)
    cur = _WS.get(key)
    if cur is None or cur.numel() < nbytes:
        _WS[key] = None
        cur = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WS[key] = cur
    return cur


# ------------------------------------------------------------------------------------------- pre-processing
def mask_minmax(mask: torch.Tensor) -> Tuple[int, int]:
    assert mask.dtype == torch.int32 and mask.is_cuda
    out = torch.empty(2, dtype=torch.int32, device=mask.device)
    check(lib().ribca_mask_minmax(ptr(mask), mask.numel(), ptr(out), stream_ptr()), "ribca_mask_minmax")
    mx, mn = out.tolist()
    return mx, mn


def label_table(mask: torch.Tensor, with_device: bool = False):
    """(H, W) int32 device mask -> (ids ascending int64 [n], table int64 [n, 7]: rmin, rmax, cmin, cmax, sum_r, sum_c,
    count) on the host.  The per-label reduction runs on the GPU; the host only drops absent labels.  ``with_device``: additionally the
    device-resident (ids int32 [n], bbox int32 [n, 4]) the crop consumes, so that a caller that needs the host table anyway (CSV
    centroids) does not send the id list back up."""
    assert mask.dtype == torch.int32 and mask.is_cuda and mask.dim() == 2
    h, w = mask.shape
    def none():
        e = (np.zeros(0, np.int64), np.zeros((0, 7), np.int64))
        return e + (torch.zeros(0, dtype=torch.int32, device=mask.device), torch.zeros((0, 4), dtype=torch.int32, device=mask.device)) if with_device else e
    if mask.numel() == 0:
        return none()
    mx, mn = mask_minmax(mask)
    if mn < 0:
        raise ValueError("segmentation mask holds negative labels; cell ids must be 1..N with 0 = background")
    if mx <= 0:
        return none()
    L = mx + 1
    ti = torch.empty((5, L), dtype=torch.int32, device=mask.device)
    tu = torch.empty((2, L), dtype=torch.int64, device=mask.device)
    check(lib().ribca_label_table(ptr(mask), h, w, L, ptr(ti), ptr(tu), stream_ptr()), "ribca_label_table")
    ti_h = ti.cpu().numpy().astype(np.int64)
    tu_h = tu.cpu().numpy()
    ids = np.flatnonzero(ti_h[4] > 0)
    table = np.stack([ti_h[0, ids], ti_h[1, ids], ti_h[2, ids], ti_h[3, ids], tu_h[0, ids], tu_h[1, ids], ti_h[4, ids]], axis=1)
    if with_device:
        ids_d = torch.nonzero(ti[4] > 0).flatten()
        return ids.astype(np.int64), table.astype(np.int64), ids_d.to(torch.int32), ti[:4].index_select(1, ids_d).t().contiguous()
    return ids.astype(np.int64), table.astype(np.int64)


def label_table_device(mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """The same reduction with the result left ON THE DEVICE: (ids int32 [n] ascending, bbox int32 [n, 4]: rmin, rmax, cmin, cmax) --
    what ``extract_patches`` consumes.  Only the label range (two scalars) and the cell count cross PCIe; the sharded path uses it so
    that no rank bounces the id list through numpy between the label table and the crop (reference preprocess.py:159-211 builds the
    whole dict on the host)."""
    assert mask.dtype == torch.int32 and mask.is_cuda and mask.dim() == 2
    h, w = mask.shape
    empty = (torch.zeros(0, dtype=torch.int32, device=mask.device), torch.zeros((0, 4), dtype=torch.int32, device=mask.device))
    if mask.numel() == 0:
        return empty
    mx, mn = mask_minmax(mask)
    if mn < 0:
        raise ValueError("segmentation mask holds negative labels; cell ids must be 1..N with 0 = background")
    if mx <= 0:
        return empty
    L = mx + 1
    ti = torch.empty((5, L), dtype=torch.int32, device=mask.device)
    tu = torch.empty((2, L), dtype=torch.int64, device=mask.device)
    check(lib().ribca_label_table(ptr(mask), h, w, L, ptr(ti), ptr(tu), stream_ptr()), "ribca_label_table")
    ids = torch.nonzero(ti[4] > 0).flatten()                       # ascending; the one host sync (its length)
    bbox = ti[:4].index_select(1, ids).t().contiguous()
    return ids.to(torch.int32), bbox


def channel_min(image: torch.Tensor) -> torch.Tensor:
    assert image.dtype == torch.float32 and image.is_cuda and image.dim() == 3
    c, h, w = image.shape
    out = torch.empty(c, dtype=torch.float32, device=image.device)
    check(lib().ribca_channel_min(ptr(image), c, h * w, ptr(out), stream_ptr()), "ribca_channel_min")
    return out


_RESIZE_PLANS: Dict[Tuple[int, int], Tuple[Optional[torch.Tensor], int, torch.Tensor]] = {}


def resize_plan(patch_size: int) -> Tuple[Optional[np.ndarray], int, np.ndarray]:
    """What ``skimage.transform.resize((C, ps, ps) -> (C, 40, 40), order=0, anti_aliasing=True)`` does per plane, as data:
    Gaussian taps at distance 0..R (sigma = (ps/40 - 1)/2 > 0 only when down-sampling; scipy's truncate-4 radius), and the 40
    nearest-neighbour source indices of ``scipy.ndimage.zoom(order=0, grid_mode=True)`` -- both computed with the same fp64
    operations as the libraries (reference call site preprocess.py:106)."""
    ps = int(patch_size)
    factor = np.divide(ps, PATCH)                              # skimage: factors = input_shape / output_shape
    sigma = max(0.0, float((factor - 1) / 2))
    taps, radius = None, 0
    if sigma > 0:
        radius = int(4.0 * sigma + 0.5)
        if radius > 0:
            taps = _gauss_weights(sigma)
    zoom = np.float64(ps) / np.float64(PATCH)                  # ndimage.zoom(grid_mode=True): input extent / output extent
    o = np.arange(PATCH, dtype=np.float64)
    cc = (o + 0.5) * zoom - 0.5                                # NI_ZoomShift: cc += 0.5; cc *= zoom; cc -= 0.5
    idx = np.floor(cc + 0.5).astype(np.int64)                  # order 0: start = floor(cc + 0.5)
    idx = np.clip(idx, 0, ps - 1)
    return taps, radius, idx.astype(np.int32)


def _resize_plan_dev(patch_size: int, device):
    key = (int(patch_size), device.index or 0)
    if key not in _RESIZE_PLANS:
        taps, radius, idx = resize_plan(patch_size)
        t = torch.from_numpy(np.ascontiguousarray(taps)).to(device) if taps is not None else None
        _RESIZE_PLANS[key] = (t, radius, torch.from_numpy(idx).to(device))
    return _RESIZE_PLANS[key]


def extract_patches(image: torch.Tensor, mask: torch.Tensor, chan_min: torch.Tensor, ids: torch.Tensor, bbox: torch.Tensor,
                    want_avg: bool = False, out: Optional[torch.Tensor] = None, patch_size: int = PATCH) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Soft-masked fp32 patches (n, C, 40, 40) for the given cells (ids int32 [n], bbox int32 [n, 4]).  ``patch_size`` =
    ``int(40 * cell_size / 30)`` (reference preprocess.py:78): windows of another size are resized to 40 x 40 as the
    reference does (preprocess.py:106)."""
    c, h, w = image.shape
    n = ids.numel()
    assert image.dtype == torch.float32 and mask.dtype == torch.int32 and ids.dtype == torch.int32 and bbox.dtype == torch.int32
    assert mask.shape == (h, w) and bbox.shape == (n, 4)
    if out is None:
        out = torch.empty((n, c, PATCH, PATCH), dtype=torch.float32, device=image.device)
    assert out.shape == (n, c, PATCH, PATCH) and out.dtype == torch.float32 and out.is_contiguous()
    avg = torch.empty((n, c), dtype=torch.float64, device=image.device) if want_avg else None
    if n and int(patch_size) == PATCH:
        check(lib().ribca_extract_patches(ptr(image), c, h, w, ptr(mask), ptr(chan_min), ptr(ids), ptr(bbox), ptr(_taps(image.device)), n,
                                          ptr(out), ptr(avg), stream_ptr()), "ribca_extract_patches")
    elif n:
        taps, radius, idx = _resize_plan_dev(patch_size, image.device)
        check(lib().ribca_extract_patches_scaled(ptr(image), c, h, w, ptr(mask), ptr(chan_min), ptr(ids), ptr(bbox), ptr(_taps(image.device)), n,
                                                 int(patch_size), ptr(taps), radius, ptr(idx), ptr(out), ptr(avg), stream_ptr()),
              "ribca_extract_patches_scaled")
    return out, avg


def colorize(mask: torch.Tensor, ids: np.ndarray, type_rgb: np.ndarray, conf_rgb: np.ndarray, type_idx: np.ndarray):
    """Paint every labelled pixel with its cell's colours (reference Annotator.colorize, model.py:806-858): ``ids`` ascending cell
    labels (n), per-cell ``type_rgb`` / ``conf_rgb`` (n, 3) uint8 and ``type_idx`` (n) uint8.  Returns device tensors (H, W, 3),
    (H, W, 3), (H, W) uint8."""
    assert mask.is_cuda and mask.dtype == torch.int32 and mask.dim() == 2
    n = len(ids)
    assert type_rgb.shape == (n, 3) and conf_rgb.shape == (n, 3) and type_idx.shape == (n,)
    h, w = mask.shape
    dev = mask.device
    top = int(ids.max()) + 1 if n else 1
    table = np.full(top, -1, dtype=np.int32)
    table[np.asarray(ids, dtype=np.int64)] = np.arange(n, dtype=np.int32)
    tab_d = torch.from_numpy(table).to(dev)
    a = torch.from_numpy(np.ascontiguousarray(type_rgb, dtype=np.uint8)).to(dev) if n else torch.zeros((1, 3), dtype=torch.uint8, device=dev)
    b = torch.from_numpy(np.ascontiguousarray(conf_rgb, dtype=np.uint8)).to(dev) if n else torch.zeros((1, 3), dtype=torch.uint8, device=dev)
    c = torch.from_numpy(np.ascontiguousarray(type_idx, dtype=np.uint8)).to(dev) if n else torch.zeros((1,), dtype=torch.uint8, device=dev)
    out_t = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    out_c = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    out_i = torch.empty((h, w), dtype=torch.uint8, device=dev)
    mask_c = mask.contiguous()
    check(lib().ribca_colorize(ptr(mask_c), h * w, ptr(tab_d), top, ptr(a), ptr(b), ptr(c), ptr(out_t), ptr(out_c), ptr(out_i),
                               stream_ptr()), "ribca_colorize")
    return out_t, out_c, out_i


def knn_cooccurrence(x: np.ndarray, y: np.ndarray, cell_type: np.ndarray, n_types: int, n_neighbors: int, out: Optional[torch.Tensor] = None,
                     device=None) -> torch.Tensor:
    """(n_types, n_types) int64 device tensor of (cell, neighbour) type pairs over each cell's n_neighbors - 1 nearest other cells
    (reference spatial_methods.neighborhood_analysis); accumulates into ``out`` when given."""
    dev = device or _lib.require_gpu()
    n = len(x)
    if n_neighbors > n:      # what scikit-learn's kneighbors raises inside the reference
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {n_neighbors}, n_samples_fit = {n}")
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(dev)
    yd = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64)).to(dev)
    td = torch.from_numpy(np.ascontiguousarray(cell_type, dtype=np.int32)).to(dev)
    if out is None:
        out = torch.zeros((n_types, n_types), dtype=torch.int64, device=dev)
    check(lib().ribca_knn_cooccurrence(ptr(xd), ptr(yd), ptr(td), n, int(n_neighbors), int(n_types), ptr(out), stream_ptr()),
          "ribca_knn_cooccurrence")
    return out


TISSUE_NEIGHBOURHOODS = (10, 20, 30, 50, 75, 100, 150, 200)       # spatial_methods.py:155


def knn_compositions(x: np.ndarray, y: np.ndarray, cell_type: np.ndarray, n_types: int, sizes: Sequence[int] = TISSUE_NEIGHBOURHOODS,
                     device=None) -> np.ndarray:
    """(n, len(sizes) * n_types) float64: for every cell and every neighbourhood size the fraction of each cell type among its
    nearest other cells -- the ``compositions`` matrix of the reference's tissue_region_partition (spatial_methods.py:158-176).
    The k-NN search and the counting run on the GPU; the division count / size is done here in fp64 as numpy does it there."""
    dev = device or _lib.require_gpu()
    n = len(x)
    if max(sizes) + 1 > n:
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {max(sizes) + 1}, n_samples_fit = {n}")
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(dev)
    yd = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64)).to(dev)
    td = torch.from_numpy(np.ascontiguousarray(cell_type, dtype=np.int32)).to(dev)
    sd = torch.tensor(list(sizes), dtype=torch.int32, device=dev)
    counts = torch.empty((n, len(sizes), n_types), dtype=torch.int16, device=dev)
    check(lib().ribca_knn_compositions(ptr(xd), ptr(yd), ptr(td), n, int(n_types), ptr(sd), len(sizes), ptr(counts), stream_ptr()),
          "ribca_knn_compositions")
    c = counts.cpu().numpy().astype(np.float64)
    c /= c.sum(axis=2, keepdims=True)
    return c.reshape(n, len(sizes) * n_types)


# ------------------------------------------------------------------------------------------- whole-image normalisation
def _gauss_weights(sigma: float) -> np.ndarray:
    """Taps at distance 0..R of scipy.ndimage.gaussian_filter(sigma, truncate=4.0), computed as scipy computes them."""
    sd = float(sigma)
    radius = int(4.0 * sd + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sd * sd) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[radius:])


def gaussian_filter_f32(x: torch.Tensor, sigma: float, tmp: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                        mode: str = "reflect") -> torch.Tensor:
    """scipy.ndimage.gaussian_filter(x, sigma) on (C, H, W) fp32 planes: axis 0 pass then axis 1 pass, bit-identical."""
    assert x.dtype == torch.float32 and x.is_cuda and x.dim() == 3
    c, h, w = x.shape
    taps = torch.from_numpy(_gauss_weights(sigma)).to(x.device)
    r = taps.numel() - 1
    tmp = torch.empty_like(x) if tmp is None else tmp
    out = torch.empty_like(x) if out is None else out
    md = {"reflect": 0, "nearest": 1}[mode]
    check(lib().ribca_gauss1d(ptr(x), ptr(tmp), c, h, w, 0, ptr(taps), r, md, stream_ptr()), "ribca_gauss1d")
    check(lib().ribca_gauss1d(ptr(tmp), ptr(out), c, h, w, 1, ptr(taps), r, md, stream_ptr()), "ribca_gauss1d")
    return out


def _order_statistics(x: torch.Tensor, ranks: np.ndarray) -> np.ndarray:
    """Exact k-th smallest value (0-based rank per plane) of non-negative fp32 planes (C, H, W) by a 3-pass radix select
    (11 + 11 + 10 key bits).  The histograms are built on the GPU; the host only walks 2048-bin prefix sums."""
    c = x.shape[0]
    hw = x.shape[1] * x.shape[2]
    dev = x.device
    hist = torch.empty((c, 2048), dtype=torch.int32, device=dev)
    prefix = np.zeros(c, dtype=np.uint32)
    remaining = np.asarray(ranks, dtype=np.int64).copy()
    mask_hi = 0
    for shift, bits in ((21, 11), (10, 11), (0, 10)):
        pre_d = torch.from_numpy(prefix.view(np.int32).copy()).to(dev)
        check(lib().ribca_radix_hist(ptr(x), c, hw, ptr(pre_d), mask_hi, shift, bits, ptr(hist), stream_ptr()), "ribca_radix_hist")
        h = hist.cpu().numpy().astype(np.int64)
        cum = np.cumsum(h, axis=1)
        for p in range(c):
            b = int(np.searchsorted(cum[p], remaining[p], side="right"))
            remaining[p] -= cum[p, b - 1] if b > 0 else 0
            prefix[p] |= np.uint32(b << shift)
        mask_hi |= ((1 << bits) - 1) << shift
    return prefix.view(np.float32).copy()


def _numpy_quantile_impl():
    """numpy's own linear-quantile helpers (index, gamma, lerp): the percentile threshold of ``_normalize`` must round exactly
    as ``np.percentile`` does.  They live in a private module that exists since NumPy 2.0 (the version this package is
    validated against); anything else fails loudly instead of silently diverging from the reference."""
    try:
        from numpy.lib import _function_base_impl as fb
        for name in ("_QuantileMethods", "_get_indexes", "_get_gamma", "_lerp"):
            getattr(fb, name)
    except (ImportError, AttributeError) as e:
        raise _lib.RibcaError(f"normalize_image needs NumPy >= 2.0 quantile internals (found numpy {np.__version__}): {e}") from e
    return fb


def _percentile_plan(n: int, amax):
    """The two sorted-array indexes np.percentile(x, amax) (method 'linear', fp32 data of n values) interpolates between,
    and its fp32 interpolation weight -- obtained by running numpy's own index/gamma helpers, so dtypes and rounding are
    numpy's whatever its version.  Returns (prev, next, gamma) with prev == next and gamma None when no interpolation."""
    fb = _numpy_quantile_impl()
    q = np.asanyarray(np.true_divide(amax, np.float32(100)))     # np.percentile divides by a.dtype.type(100) for float data
    virtual = np.asanyarray(fb._QuantileMethods["linear"]["get_virtual_index"](n, q))
    if np.issubdtype(virtual.dtype, np.integer):
        return int(virtual), int(virtual), None
    prev, nxt = fb._get_indexes(np.empty((0,), dtype=np.float32), virtual, n)
    gamma = fb._get_gamma(virtual, prev, fb._QuantileMethods["linear"])
    return int(prev) % n, int(nxt) % n, gamma


def _percentile_like_numpy(n: int, amax, lo_hi_getter):
    """np.percentile of one fp32 plane given a callable returning its exact order statistics (prev, next)."""
    fb = _numpy_quantile_impl()
    prev, nxt, gamma = _percentile_plan(n, amax)
    a, b = lo_hi_getter(np.array([prev]), np.array([nxt]))
    return np.float32(a) if gamma is None else fb._lerp(np.float32(a), np.float32(b), gamma)


def normalize_image(raw, blur=0, amax=100, u16_bits: bool = False) -> torch.Tensor:
    """ImageProcessor._normalize (reference preprocess.py:214-239) on the GPU, bit-identical to the CPU path.
    ``raw``: (C, H, W) numpy array or torch tensor of any real dtype; returns a fp32 CUDA tensor in [-1, 1].
    ``u16_bits``: a torch.int16 tensor holds uint16 pixel BITS (how a device-resident uint16 image is carried, torch having no
    uint16 arithmetic); without the flag int16 data is signed and converted by value like every other dtype."""
    fb = _numpy_quantile_impl()
    dev = _lib.require_gpu()
    if isinstance(raw, np.ndarray):
        if raw.dtype == np.uint16:
            src = torch.from_numpy(np.ascontiguousarray(raw).view(np.int16)).to(dev)
            img = torch.empty(raw.shape, dtype=torch.float32, device=dev)
            check(lib().ribca_u16_to_f32(ptr(src), ptr(img), src.numel(), stream_ptr()), "ribca_u16_to_f32")
        else:
            img = torch.from_numpy(raw.astype(np.float32)).to(dev)
    elif raw.dtype == torch.uint16 or (raw.dtype == torch.int16 and u16_bits):     # uint16 pixels already resident on the device
        src = raw.to(dev).contiguous()
        if src.dtype == torch.uint16:
            src = src.view(torch.int16)
        img = torch.empty(tuple(raw.shape), dtype=torch.float32, device=dev)
        check(lib().ribca_u16_to_f32(ptr(src), ptr(img), src.numel(), stream_ptr()), "ribca_u16_to_f32")
    else:
        img = raw.to(device=dev, dtype=torch.float32).contiguous().clone()
    c, h, w = img.shape
    hw = h * w
    tmp = torch.empty_like(img)
    bg = gaussian_filter_f32(img, 20, tmp=tmp)
    check(lib().ribca_bg_subtract(ptr(img), ptr(bg), img.numel(), 125.0, stream_ptr()), "ribca_bg_subtract")
    if blur:
        blurred = gaussian_filter_f32(img, blur, tmp=tmp, out=bg)
        img, bg = blurred, img
    mx_d = torch.empty(c, dtype=torch.float32, device=dev)
    check(lib().ribca_plane_max(ptr(img), c, hw, ptr(mx_d), stream_ptr()), "ribca_plane_max")
    mx = mx_d.cpu().numpy()
    mode = np.zeros(c, np.int32)                 # 0: no positive pixel -> plane of -1
    clip = np.full(c, np.inf, np.float32)
    denom = np.ones(c, np.float32)
    sel = np.flatnonzero(mx > 0)
    if sel.size:
        prev, nxt, gamma = _percentile_plan(hw, amax)
        planes = img[torch.from_numpy(sel).to(dev)] if sel.size < c else img
        lo = _order_statistics(planes, np.full(sel.size, prev))
        hi = lo if nxt == prev else _order_statistics(planes, np.full(sel.size, nxt))
        for j, p in enumerate(sel):
            t = np.float32(lo[j]) if gamma is None else fb._lerp(np.float32(lo[j]), np.float32(hi[j]), gamma)
            m = np.float32(mx[p])
            if t > 20:                            # preprocess.py:234-235
                clip[p] = t
                m = min(m, np.float32(t))
            mode[p] = 1
            denom[p] = max(25, m)                 # preprocess.py:238
    # one named tensor per argument: a temporary would be returned to the caching allocator (and its block handed to the next
    # .to(dev)) before the kernel that reads it is enqueued
    mode_d = torch.from_numpy(mode).to(dev)
    clip_d = torch.from_numpy(clip).to(dev)
    denom_d = torch.from_numpy(denom).to(dev)
    check(lib().ribca_norm_finalize(ptr(img), c, hw, ptr(mode_d), ptr(clip_d), ptr(denom_d), stream_ptr()), "ribca_norm_finalize")
    return img


# ------------------------------------------------------------------------------------------- ViT
def decision_distance(pa: torch.Tensor, others_a: Optional[int], pb: Optional[torch.Tensor], others_b: Optional[int],
                      thresholds: Sequence[float]) -> torch.Tensor:
    """How far (in probability) each cell is from the nearest boundary of the vote (reference model.py:481-633): the smaller of the top-2
    margin among the candidate classes and the distance of the best candidate from every quantity it is compared with -- the models'
    "Others" probabilities, the confidence threshold, the per-type thresholds in force.  Conservative (a threshold that does not bind
    for a cell still counts): it decides which cells Annotator.predict re-evaluates at full precision and which it reports as inside the
    arithmetic's noise floor.  ``others_x`` = column of "Others" in table x (None: the table has none)."""
    def split(p, o):
        if o is None:
            return p, None
        keep = [i for i in range(p.shape[1]) if i != o]
        return p[:, keep], p[:, o]
    va, oa = split(pa, others_a)
    cand, oth = [va], [oa] if oa is not None else []
    if pb is not None:
        vb, ob = split(pb, others_b)
        cand.append(vb)
        if ob is not None:
            oth.append(ob)
    v = torch.cat(cand, dim=1)
    top = torch.topk(v, min(2, v.shape[1]), dim=1).values
    best = top[:, 0]
    dist_ = (top[:, 0] - top[:, -1]) if v.shape[1] > 1 else torch.full_like(best, 2.0)
    for o in oth:
        dist_ = torch.minimum(dist_, (best - o).abs())
    for t in thresholds:
        if t is not None and t >= 0:
            dist_ = torch.minimum(dist_, (best - float(t)).abs())
    return dist_


class VitModel:
    """One packed classifier on one device (replaces a timm ``VisionTransformer`` instance of reference
    model.py:188-234).  ``state_dict`` uses the timm key names of the reference checkpoints."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device=None):
        device = device or _lib.require_gpu()
        self.device = torch.device(device)
        d = state_dict["cls_token"].shape[-1]
        c = state_dict["patch_embed.proj.weight"].shape[1]
        k = state_dict["head.weight"].shape[0]
        depth = 0
        while f"blocks.{depth}.norm1.weight" in state_dict:
            depth += 1
        if state_dict["pos_embed"].shape[1] != 101 or tuple(state_dict["patch_embed.proj.weight"].shape[2:]) != (4, 4):
            raise ValueError("expected a 40x40 / patch-4 ViT (101 tokens)")
        self.D, self.C, self.K, self.depth = int(d), int(c), int(k), depth
        blob = torch.cat([state_dict[key].detach().to(torch.float32).reshape(-1) for key in blob_keys(depth)]).to(self.device)
        want = lib().ribca_vit_blob_len(self.D, self.C, self.K, depth)
        if blob.numel() != want:
            raise ValueError(f"state dict has {blob.numel()} parameters, expected {want}")
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().ribca_vit_create(ptr(blob), blob.numel(), self.D, self.C, self.K, depth, stream_ptr(), ctypes.byref(handle)),
                  "ribca_vit_create")
            torch.cuda.current_stream().synchronize()  # the blob may be freed once packing has finished
        self._h = handle
        self.flops_per_cell = float(lib().ribca_vit_flops_per_cell(self._h))
        self.probe_fast_minus_full = 0.0
        self.probe_logit_delta = 0.0
        self.probe_predicted_dp = 0.0
        self.fast_ok = True
        if lib().ribca_mx_enabled(self.D) and os.environ.get("RIBCA_MARGIN_PROBE", "1") != "0":
            self._calibrate_margin()

    # ---- whether a model's weights tolerate the MX arithmetic is measured, not assumed (round 6) ---------------------------------------
    PROBE_CELLS = 64
    #: largest |dp| of a real cell over 0.25 x the probe's logit-difference move: <= 2.75 in 64 (family, head, seed, classifier) cases over
    #: 6000 real cells each (profiles/r6/probe_vs_real.txt); 3 is what the rule assumes
    PROBE_FACTOR = 3.0
    #: the fast (MX) forward is used only where the predicted worst |fast - full precision| stays below RECHECK_MARGIN / PROBE_DIVISOR
    PROBE_DIVISOR = 2.5

    def _calibrate_margin(self) -> None:
        """What the margin-gated re-evaluation rests on is |fast - full precision| staying well inside RECHECK_MARGIN for every cell (a cell
        the fast path places outside the margin then cannot cross a boundary at full precision), and what the 1e-3 confidence tolerance
        rests on is the same distance staying a fraction of it.  On the uniform synthetic family with the audit's soft head the distance is
        1-3e-5 over 2000 real cells; a sharper head (the bench's head_gain 4) or weights with heavy tails, LayerNorm gains two decades apart
        and massive-activation channels -- what trained ViTs have (synth.make_vit_state_dict_heavy) -- move the block-scaled correction
        products by 1-6e-4.  So it is MEASURED once per model, at load time, on a fixed synthetic probe (PROBE_CELLS patches: background
        -1, sparse positive signal; the same cells whatever the image, the rank or the chunk, so a cell's treatment never depends on
        where it was computed).  The statistic is saturation-free: delta = the largest move of a LOGIT DIFFERENCE, log p_i - log p_top,
        between the two forwards (a probe cell whose softmax is saturated hides any move of its probabilities: the probability form of
        this probe under-predicted real cells by up to 100 x); a probability then moves by at most p (1 - p) delta <= delta / 4.  Over 64
        cases (two weight families x two head gains x four seeds x the four MX-capable classifiers) the largest |dp| of 6000 real cells
        was 0.18 ... 2.75 x delta / 4 (tools/probe_vs_real.py, profiles/r6/probe_vs_real.txt), so the predicted worst case is
        PROBE_FACTOR x delta / 4 and a model is accepted while that stays below RECHECK_MARGIN / 2.5 = 4e-4 (delta <= 5.3e-4): every
        accepted case measured <= 2.5e-4.  A model beyond the bar runs EVERY product at three fp16 passes (``fast_ok`` False: the
        precise forward for all cells, slower, nothing to re-evaluate).  Costs two 64-cell forwards per model."""
        g = torch.Generator().manual_seed(0x5249424341)
        u = torch.rand((self.PROBE_CELLS, self.C, PATCH, PATCH), generator=g, dtype=torch.float32) * 2.0 - 1.0
        x = torch.where(u > 0.1, u, torch.full_like(u, -1.0)).to(self.device)
        with torch.cuda.device(self.device):
            src = list(range(self.C))
            fast = self._forward(x, src, chunk_cells=self.PROBE_CELLS, precise=False).double()
            full = self._forward(x, src, chunk_cells=self.PROBE_CELLS, precise=True).double()
            self.probe_fast_minus_full = float((fast - full).abs().max().item())       # (informational: the probability form)
            ok = (fast > 1e-30) & (full > 1e-30)
            lf, lp = torch.log(fast.clamp_min(1e-300)), torch.log(full.clamp_min(1e-300))
            top = full.argmax(1, keepdim=True)
            d = ((lf - lf.gather(1, top)) - (lp - lp.gather(1, top))).abs()
            self.probe_logit_delta = float(d[ok].max().item()) if bool(ok.any()) else 0.0
        self.probe_predicted_dp = self.PROBE_FACTOR * 0.25 * self.probe_logit_delta
        self.fast_ok = self.probe_predicted_dp <= float(type(self).RECHECK_MARGIN) / self.PROBE_DIVISOR

    @property
    def recheck_margin(self) -> float:
        """cells whose fast result lies this close to a decision boundary are re-evaluated at full operand precision"""
        return float(type(self).RECHECK_MARGIN)

    @property
    def uses_mx(self) -> bool:
        """True where this model's forward really runs the MX products (the width allows them AND the probe accepted the weights)"""
        return bool(lib().ribca_mx_enabled(self.D)) and self.fast_ok

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().ribca_vit_destroy(h)
            except Exception:
                pass
            self._h = None

    # Cells whose fast (MX) result lies this close to a decision boundary are re-evaluated with three fp16 passes per product.  The MX
    # pair moves softmax outputs by 2-4e-5 against the fp16x3 path (measured; 1e-4 class with every product in that form, emulated):
    # 1e-3 leaves an order of magnitude, and is the north star's own confidence tolerance.  It is the FLOOR of the margin: the margin a
    # model actually uses is ``recheck_margin`` (calibrated on the model's own weights at load time, _calibrate_margin).
    RECHECK_MARGIN = 1.0e-3

    #: width -> multiple of the caller's chunk_cells a forward of that width uses (RIBCA_CHUNK_SCALE=<n> overrides it for every width: A/B)
    CHUNK_SCALE = {576: 4}

    #: the scaled chunk never exceeds this many cells (the workspace of a 576-wide forward is ~ 3.9 MB per cell and there is one per segment
    #: stream: 4096 cells x 3 streams = 48 GB of the 288; a caller that asks for more than this itself gets what it asked for, unscaled)
    MAX_SCALED_CHUNK = 4096

    def chunk_scale(self) -> int:
        env = os.environ.get("RIBCA_CHUNK_SCALE")
        return max(1, int(env)) if env else self.CHUNK_SCALE.get(self.D, 1)

    def effective_chunk(self, chunk_cells: int) -> int:
        """cells per launch sequence a forward of this width really uses for a caller's ``chunk_cells`` (bench.py reports it per model)"""
        c = max(1, int(chunk_cells))
        return max(c, min(c * self.chunk_scale(), self.MAX_SCALED_CHUNK))

    def predict_proba(self, patches: torch.Tensor, src_chan: Sequence[int], chunk_cells: int = 1024, ws_slot: int = 0,
                      streams: int = 1, recheck: Optional[Sequence[float]] = None) -> torch.Tensor:
        """``_forward`` plus, where this width runs the MX pair, the full-precision re-evaluation of the cells near a decision boundary:
        ``recheck`` = the thresholds the caller will compare confidences with (``[]``: only the top-2 margin counts; ``None``: no
        re-evaluation).  A cell is re-evaluated when its two best classes are closer than RECHECK_MARGIN or its best probability is
        within RECHECK_MARGIN of a threshold -- a property of the cell's own fast result, so results stay independent of chunking.
        ``self.last_recheck`` = {"cells": re-evaluated, "undecidable": of those, still within 2e-4 of a boundary afterwards}."""
        probs = self._forward(patches, src_chan, chunk_cells, ws_slot, streams, precise=False)
        self.last_recheck = {"cells": 0, "undecidable": 0}
        if recheck is None or probs.shape[0] == 0 or not self.uses_mx:      # (every product already at three fp16 passes: nothing to re-evaluate)
            return probs

        def near(p, eps):
            top = torch.topk(p, min(2, p.shape[1]), dim=1).values
            m = (top[:, 0] - top[:, -1]) < eps if p.shape[1] > 1 else torch.zeros(p.shape[0], dtype=torch.bool, device=p.device)
            for t in recheck:
                if t is not None and t > 0:
                    m |= (top[:, 0] - float(t)).abs() < eps
            return m

        idx = torch.nonzero(near(probs, self.recheck_margin)).flatten()
        if idx.numel():
            again = self._forward(patches.index_select(0, idx), src_chan, chunk_cells, ws_slot, 1, precise=True)
            probs.index_copy_(0, idx, again)
            self.last_recheck = {"cells": int(idx.numel()), "undecidable": int(near(again, 2.0e-4).sum().item())}
        return probs

    def _forward(self, patches: torch.Tensor, src_chan: Sequence[int], chunk_cells: int = 1024, ws_slot: int = 0,
                 streams: int = 1, precise: bool = False, force_fast: bool = False) -> torch.Tensor:
        """softmax(model(x), dim=1) for full-channel patches (n, C_img, 40, 40); ``src_chan[c]`` = image channel of model
        channel c or -1 for a blank plane (reference preprocess.py:110-120, model.py:397-406).

        ``streams`` > 1 splits the cells into that many contiguous segments and enqueues each on its own HIP stream (own
        workspace slot): cells are independent, so one segment's launch gaps, tile tails and store bursts overlap another
        segment's MFMA work.  The result is identical to the single-stream run (every cell sees the same arithmetic)."""
        assert patches.is_cuda and patches.dtype == torch.float32 and patches.dim() == 4 and patches.shape[2:] == (PATCH, PATCH)
        if len(src_chan) != self.C:
            raise ValueError(f"model expects {self.C} channels, got an index list of {len(src_chan)}")
        n, c_img = patches.shape[0], patches.shape[1]
        if any(s >= c_img or s < -1 for s in src_chan):
            raise ValueError("channel index out of range")
        probs = torch.empty((n, self.K), dtype=torch.float32, device=patches.device)
        if n == 0:
            return probs
        # force_fast: the MX products even where the probe refused this model's weights (tests and audits measure what the rule protects from)
        fwd = lib().ribca_vit_forward_precise if (precise or not (self.fast_ok or force_fast)) else lib().ribca_vit_forward
        # callers size chunks for the ensemble (1024 cells = 103 424 GEMM rows); the 576-wide classifier's launches are then 4.7 rounds of
        # workgroups and lose a twentieth each to the partial last round plus a fixed 39 us of ramp (profiles/r5/mx_rounds.txt): it takes
        # CHUNK_SCALE times the caller's chunk (same-box sweep, profiles/r5/chunk_streams_full.txt: 39.3-39.9 -> 40.4-41.3 k cells/s at 4 x;
        # the narrower classifiers measured flat or slightly worse).  Results do not depend on the chunk size (test_classifier_bitwise_repeatable).
        chunk = max(1, min(self.effective_chunk(chunk_cells), n))
        src = torch.tensor(list(src_chan), dtype=torch.int32, device=patches.device)
        nbytes = lib().ribca_vit_workspace_bytes(self._h, chunk)
        patches = patches.contiguous()
        n_chunks = (n + chunk - 1) // chunk
        nseg = max(1, min(int(streams), n_chunks))
        if nseg == 1:
            ws = workspace(nbytes + 256, patches.device, ws_slot)
            aligned = (ws.data_ptr() + 255) & ~255
            check(fwd(self._h, ptr(patches), c_img, ptr(src), n, ptr(probs), aligned, nbytes, chunk, stream_ptr()), "ribca_vit_forward")
            return probs
        main = torch.cuda.current_stream(patches.device)
        ready = torch.cuda.Event()
        ready.record(main)
        side = _side_streams(patches.device, nseg)
        for i, st in enumerate(side):
            c0, c1 = n_chunks * i // nseg, n_chunks * (i + 1) // nseg
            lo, hi = c0 * chunk, min(c1 * chunk, n)
            if hi <= lo:
                continue
            ws = workspace(nbytes + 256, patches.device, (ws_slot + 1) * 64 + i)
            aligned = (ws.data_ptr() + 255) & ~255
            st.wait_event(ready)
            with torch.cuda.stream(st):
                check(fwd(self._h, ptr(patches[lo:hi]), c_img, ptr(src), hi - lo, ptr(probs[lo:hi]), aligned, nbytes, chunk, stream_ptr()),
                      "ribca_vit_forward")
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)
        return probs


_SIDE: Dict[Tuple[int, int], "torch.cuda.Stream"] = {}


def _side_streams(device, n: int):
    out = []
    for i in range(n):
        key = (device.index or 0, i)
        if key not in _SIDE:
            _SIDE[key] = torch.cuda.Stream(device=device)
        out.append(_SIDE[key])
    return out


def mae_blob_keys(enc_depth: int, dec_depth: int) -> List[str]:
    def blk(prefix, depth):
        out = []
        for i in range(depth):
            p = f"{prefix}{i}."
            out += [p + "norm1.weight", p + "norm1.bias", p + "attn.qkv.weight", p + "attn.qkv.bias", p + "attn.proj.weight",
                    p + "attn.proj.bias", p + "norm2.weight", p + "norm2.bias", p + "mlp.fc1.weight", p + "mlp.fc1.bias",
                    p + "mlp.fc2.weight", p + "mlp.fc2.bias"]
        return out
    return (["cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"] + blk("blocks.", enc_depth)
            + ["norm.weight", "norm.bias", "decoder_embed.weight", "decoder_embed.bias", "mask_token", "decoder_pos_embed"]
            + blk("decoder_blocks.", dec_depth) + ["decoder_norm.weight", "decoder_norm.bias", "decoder_pred.weight", "decoder_pred.bias"])


class MaeModel:
    """Packed marker imputer on one device (replaces the ``MaskedAutoencoderViT`` inside the reference's
    ``MarkerImputer``, markerImputer.py:258-287); ``state_dict`` = the ``["model"]`` entry of a ``*_impute.pth`` checkpoint."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device=None):
        device = device or _lib.require_gpu()
        self.device = torch.device(device)
        self.L = int(state_dict["pos_embed"].shape[1]) - 1
        enc = dec = 0
        while f"blocks.{enc}.norm1.weight" in state_dict:
            enc += 1
        while f"decoder_blocks.{dec}.norm1.weight" in state_dict:
            dec += 1
        if state_dict["cls_token"].shape[-1] != 768 or state_dict["mask_token"].shape[-1] != 512 or \
                tuple(state_dict["patch_embed.proj.weight"].shape) != (768, 1, 40, 40):
            raise ValueError("expected the reference imputer geometry (768/512 wide, one 40x40 tile per token)")
        blob = torch.cat([state_dict[k].detach().to(torch.float32).reshape(-1) for k in mae_blob_keys(enc, dec)]).to(self.device)
        if blob.numel() != lib().ribca_mae_blob_len(self.L, enc, dec):
            raise ValueError("imputer state dict does not match (L, depths)")
        self._h = self._create(blob, enc, dec)
        self.probe_plane_delta = 0.0
        self.fast_ok = True
        # the imputer's fast path (folded blocks, 768-wide encoder on the MX kernel) has no full-precision twin inside one handle: where the
        # environment leaves the choice open, a second handle on the round-2 path (three fp16 passes everywhere) is the yardstick of a load-time
        # probe, as for the classifiers (VitModel._calibrate_margin)
        if os.environ.get("RIBCA_MARGIN_PROBE", "1") != "0" and os.environ.get("RIBCA_MAE_FOLD") is None:
            self._probe(blob, enc, dec)

    def _create(self, blob, enc, dec):
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().ribca_mae_create(ptr(blob), blob.numel(), self.L, enc, dec, stream_ptr(), ctypes.byref(handle)), "ribca_mae_create")
            torch.cuda.current_stream().synchronize()
        return handle

    #: the fast imputer is kept while its imputed pixels (values in [-1, 1]) stay this close to the fp16x3 imputer's on the probe: the uniform
    #: family measures 8e-5 on real cells, which moves the classifier's confidences behind it by 2e-6 (tests/test_gpu_e2e.py, config-5 audit)
    PROBE_LIMIT = 1.0e-3
    PROBE_CELLS = 64

    def _probe(self, blob, enc, dec) -> None:
        os.environ["RIBCA_MAE_FOLD"] = "0"          # read by the library at create
        try:
            slow = self._create(blob, enc, dec)
        finally:
            del os.environ["RIBCA_MAE_FOLD"]
        g = torch.Generator().manual_seed(0x5249424342)
        u = torch.rand((self.PROBE_CELLS, self.L, PATCH, PATCH), generator=g, dtype=torch.float32) * 2.0 - 1.0
        x = torch.where(u > 0.1, u, torch.full_like(u, -1.0))
        x[:, self.L - 1] = -1.0                      # the last marker missing, as in BASELINE config 5
        present = list(range(self.L - 1))
        a, b = x.to(self.device).contiguous(), x.to(self.device).contiguous()
        with torch.cuda.device(self.device):
            self._impute_with(self._h, a, present, self.PROBE_CELLS)
            self._impute_with(slow, b, present, self.PROBE_CELLS)
            self.probe_plane_delta = float((a[:, self.L - 1] - b[:, self.L - 1]).abs().max().item())
        self.fast_ok = self.probe_plane_delta <= self.PROBE_LIMIT
        drop = slow if self.fast_ok else self._h
        if not self.fast_ok:
            self._h = slow                           # these weights do not tolerate the MX products: every product at three fp16 passes
        lib().ribca_mae_destroy(drop)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().ribca_mae_destroy(h)
            except Exception:
                pass
            self._h = None

    #: a cell is <= 16 token rows here against 101 in a classifier: callers that size chunks for the classifiers (Annotator, bench.py) hand the
    #: imputer this many times their chunk, so that its GEMMs see a comparable row count (50 000 cells: 0.489 s at 1024, 0.464 s at 4096;
    #: profiles/r5/ab_mae_fold.txt).  Results do not depend on the chunk size (tests/test_gpu_e2e.py::test_config5_full_size_properties).
    CHUNK_FACTOR = 4

    def impute(self, patches: torch.Tensor, present: Sequence[int], chunk_cells: int = 1024) -> torch.Tensor:
        """In place: every channel of ``patches`` (n, L, 40, 40) not listed in ``present`` is replaced by its prediction."""
        return self._impute_with(self._h, patches, present, chunk_cells)

    def _impute_with(self, handle, patches: torch.Tensor, present: Sequence[int], chunk_cells: int) -> torch.Tensor:
        assert patches.is_cuda and patches.dtype == torch.float32 and patches.is_contiguous() and tuple(patches.shape[1:]) == (self.L, PATCH, PATCH)
        n = patches.shape[0]
        pres = sorted(int(c) for c in present)
        if n == 0:
            return patches
        chunk = max(1, min(int(chunk_cells), n))
        nbytes = lib().ribca_mae_workspace_bytes(handle, chunk, len(pres))
        if nbytes <= 0:
            raise ValueError("need at least one present and one missing channel")
        ws = workspace(nbytes + 256, patches.device)
        aligned = (ws.data_ptr() + 255) & ~255
        arr = (ctypes.c_int32 * len(pres))(*pres)
        check(lib().ribca_mae_impute(handle, ptr(patches), arr, len(pres), n, aligned, nbytes, chunk, stream_ptr()), "ribca_mae_impute")
        return patches


def resolve_channels(channel_index: Sequence[int], c_img: int) -> List[int]:
    """Reference channel-select semantics (preprocess.py:110-120): the FIRST -1 becomes a blank plane, every further -1
    stays in a numpy fancy index and therefore picks the LAST image channel."""
    out, seen_blank = [], False
    for ci in channel_index:
        if ci == -1 and not seen_blank:
            out.append(-1)
            seen_blank = True
        elif ci == -1:
            out.append(c_img - 1)
        else:
            out.append(int(ci))
    return out


# ------------------------------------------------------------------------------------------- vote
def vote(p_a: torch.Tensor, map_a: Sequence[int], p_b: Optional[torch.Tensor], map_b: Optional[Sequence[int]],
         type_conf: Sequence[float], conf: float) -> Tuple[torch.Tensor, torch.Tensor]:
    n = p_a.shape[0]
    dev = p_a.device
    label = torch.empty(n, dtype=torch.int8, device=dev)
    out_conf = torch.empty(n, dtype=torch.float32, device=dev)
    ma = torch.tensor(list(map_a), dtype=torch.int8, device=dev)
    mb = torch.tensor(list(map_b), dtype=torch.int8, device=dev) if p_b is not None else None
    tc = torch.tensor([float(np.float32(v)) for v in type_conf], dtype=torch.float32, device=dev)
    pa_c = p_a.contiguous()                                   # named: only tensors that outlive the call cross the ABI
    pb_c = p_b.contiguous() if p_b is not None else None
    check(lib().ribca_vote(ptr(pa_c), pa_c.shape[1], ptr(ma), ptr(pb_c) if pb_c is not None else None,
                           pb_c.shape[1] if pb_c is not None else 0, ptr(mb), ptr(tc), float(np.float32(conf)), n, ptr(label),
                           ptr(out_conf), stream_ptr()), "ribca_vote")
    return label, out_conf


# ------------------------------------------------------------------------------------------- profiling
def prof_enable(on: bool) -> None:
    lib().ribca_prof_enable(1 if on else 0)


def prof_read() -> Dict[str, Tuple[float, int]]:
    ms = (ctypes.c_double * 10)()
    cnt = (ctypes.c_int64 * 10)()
    lib().ribca_prof_read(ms, cnt)
    return {lib().ribca_prof_name(i).decode(): (ms[i], cnt[i]) for i in range(10)}
