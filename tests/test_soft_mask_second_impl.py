"""Second, independent implementation of the reference's soft mask (``utils.smooth``, cell_type_annotation/utils.py:255-270) against
the oracle's (oracle/ref_preprocess.soft_mask on oracle/skimage_like.py).

scikit-image is not installed and not vendored, so the goldens' ``dilation(disk(j))`` / ``filters.gaussian`` ran through the
builder's restatement on scipy.ndimage (VERDICT r2 missing #3: the only third-party arithmetic with neither the real library nor
an independent cross-check).  This file restates the PUBLISHED semantics a second time without scipy.ndimage's morphology or
filters:

* ``dilation(m, disk(j))``  = {p : min over q in m of |p - q|^2 <= j^2}  -- exact integer squared distances, brute force
  (and, as a third opinion, ``scipy.ndimage.distance_transform_edt(~m) <= j``);
* ``filters.gaussian(d, sigma)`` = separable correlation of the float64 0/1 image with w[k] = exp(-k^2 / 2 sigma^2) / sum,
  k = -r..r, r = int(4 sigma + 0.5), edge replication ('nearest'), axis 0 then axis 1 -- plain numpy sums;
* the reference's own accumulation: fp32 S = m; for j = 1..4: S += d_j; for s = 1..j-1: S += G_s(d_j); S /= 11; S /= max(S + 1e-6).
"""
import os

import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle import ref_preprocess as rp
from oracle import skimage_like as ski


def dilate_bruteforce(m: np.ndarray, j: int) -> np.ndarray:
    ys, xs = np.nonzero(m)
    if len(ys) == 0:
        return np.zeros_like(m, dtype=bool)
    yy, xx = np.mgrid[0:m.shape[0], 0:m.shape[1]]
    d2 = (yy[..., None] - ys) ** 2 + (xx[..., None] - xs) ** 2          # integers: exact
    return d2.min(axis=-1) <= j * j


def gaussian_explicit(img01: np.ndarray, sigma: float) -> np.ndarray:
    r = int(4.0 * sigma + 0.5)
    k = np.arange(-r, r + 1, dtype=np.float64)
    w = np.exp(-0.5 * k * k / (sigma * sigma))
    w /= w.sum()
    x = img01.astype(np.float64)
    for axis in (0, 1):
        pad = [(0, 0), (0, 0)]
        pad[axis] = (r, r)
        xp = np.pad(x, pad, mode="edge")
        out = np.zeros_like(x)
        for t in range(2 * r + 1):
            sl = [slice(None), slice(None)]
            sl[axis] = slice(t, t + x.shape[axis])
            out += w[t] * xp[tuple(sl)]
        x = out
    return x


def soft_mask_second(mask_patch: np.ndarray, cell_id) -> np.ndarray:
    m = mask_patch == cell_id
    s = m.astype(np.float32)
    count = 1
    for j in range(1, 5):
        d = dilate_bruteforce(m, j)
        s += d
        count += 1
        for i in range(j - 1):
            s += gaussian_explicit(d, 1 + i)
            count += 1
    s /= count
    s /= np.max(s + 1e-6)
    return s


def _windows(mask, ps=40):
    ids, tab = rp.cell_table(mask)
    for cid, row in zip(ids, tab):
        r0, r1, c0, c1 = rp.window_bounds(int(row[0]), int(row[1]), int(row[2]), int(row[3]), mask.shape[0], mask.shape[1], ps)
        win = np.zeros((ps, ps), dtype=mask.dtype)
        win[:r1 - r0, :c1 - c0] = mask[r0:r1, c0:c1]
        yield int(cid), win


@pytest.fixture(scope="module")
def golden_masks(golden_dir):
    g = np.load(os.path.join(golden_dir, "patches.npz"))
    cp = np.load(os.path.join(golden_dir, "cellpos.npz"))
    masks = [g["A_mask"], g["B_mask"]]
    if "example2_mask" in cp.files:
        masks.append(cp["example2_mask"][:200, :200])          # the reference's own example mask (a corner: the brute force is O(n^2))
    return masks


def test_pieces_agree(golden_masks):
    """dilations identical (three ways), Gaussians to 1e-12"""
    n = 0
    for mask in golden_masks:
        for cid, win in _windows(mask):
            m = win == cid
            for j in range(1, 5):
                d_oracle = ski.dilation(m, ski.disk(j))
                d_mine = dilate_bruteforce(m, j)
                assert np.array_equal(d_oracle, d_mine)
                if m.any():
                    assert np.array_equal(ndi.distance_transform_edt(~m) <= j, d_mine)
                for sigma in range(1, j):
                    a, b = ski.gaussian(d_oracle, sigma=sigma), gaussian_explicit(d_mine, sigma)
                    assert np.abs(a - b).max() <= 1e-12
            n += 1
    assert n >= 40


def test_soft_mask_agrees(golden_masks):
    """the fp32 result: the two Gaussians differ by ~1e-16 (summation order), which can move an fp32 accumulation by one ulp at most"""
    worst, exact, total = 0.0, 0, 0
    for mask in golden_masks:
        for cid, win in _windows(mask):
            a, b = rp.soft_mask(win, cid), soft_mask_second(win, cid)
            assert a.dtype == np.float32 and b.dtype == np.float32
            worst = max(worst, float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()))
            exact += int(np.array_equal(a, b))
            total += 1
    assert worst <= 1.2e-7, worst          # one fp32 ulp at 1.0
    assert exact >= 0.9 * total, (exact, total)


def test_golden_smooth_entries(golden_dir):
    """the two windows the REFERENCE's own utils.smooth was run on for the fixture (tests/golden/make_golden.py: mask A rows 20..59,
    columns 30..69, cell ids 65000 and 41; through the skimage stand-in): the second implementation lands on the same fp32 values"""
    g = np.load(os.path.join(golden_dir, "patches.npz"))
    lab = np.zeros((40, 40))
    lab[:40, :40] = g["A_mask"][20:60, 30:70]
    for cid, key in ((65000, "A_smooth_big"), (41, "A_smooth_nested")):
        got = soft_mask_second(lab, cid)
        assert np.abs(got.astype(np.float64) - g[key].astype(np.float64)).max() <= 1.2e-7
