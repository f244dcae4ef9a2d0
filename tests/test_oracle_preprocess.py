"""Oracle (CPU restatement) vs golden vectors produced by the reference's own functions (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import ref_preprocess as rp


@pytest.fixture(scope="module")
def g_norm(golden_dir):
    return np.load(os.path.join(golden_dir, "normalize.npz"))


@pytest.mark.parametrize("key,src,blur,amax", [
    ("a_out_blur0", "a_in", 0, 99.8), ("a_out_blur0.3", "a_in", 0.3, 99.8), ("a_out_blur0.5", "a_in", 0.5, 99.8),
    ("a_out_blur1", "a_in", 1, 99.8), ("a_out_amax100", "a_in", 0, 100), ("b_out_blur0.3", "b_in", 0.3, 99.8),
    ("c_out_blur0", "c_in", 0, 99.8)])
def test_normalize_matches_reference(g_norm, key, src, blur, amax):
    out = rp.normalize_image(g_norm[src], blur=blur, amax=amax)
    assert out.dtype == np.float32
    np.testing.assert_array_equal(out, g_norm[key])  # same numpy/scipy calls -> bit-exact


def test_normalize_degenerate_channels(g_norm):
    out = g_norm["a_out_blur0"]
    assert (out[2] == -1).all()            # channel without positive pixels
    assert out[3].max() < 1.0              # max < 25 -> scaled by 25, never reaches +1
    assert out[0].max() == 1.0 and out[0].min() == -1.0


@pytest.fixture(scope="module")
def g_cell(golden_dir):
    return np.load(os.path.join(golden_dir, "cellpos.npz"))


def test_cell_table_odd_mask(g_cell):
    ids, tab = rp.cell_table(g_cell["odd_mask"])
    np.testing.assert_array_equal(ids, g_cell["odd_ids"])
    np.testing.assert_array_equal(tab, g_cell["odd_table"])
    assert ids.tolist() == sorted(ids.tolist()) and 100000 in ids and 0 not in ids


def test_cell_table_example_mask(g_cell):
    ids, tab = rp.cell_table(g_cell["example2_mask"].astype(np.int32))
    assert len(ids) == 582
    np.testing.assert_array_equal(ids, g_cell["example2_ids"])
    np.testing.assert_array_equal(tab, g_cell["example2_table"])


def test_cell_table_example1_mask(g_cell):
    """BASELINE config 1's mask (examples/example_1_cell_mask.png, 1850 cells)."""
    ids, tab = rp.cell_table(g_cell["example1_mask"].astype(np.int32))
    assert len(ids) == 1850
    np.testing.assert_array_equal(ids, g_cell["example1_ids"])
    np.testing.assert_array_equal(tab, g_cell["example1_table"])


def test_cell_positions_scan_order(g_cell):
    d = rp.cell_positions(g_cell["odd_mask"])
    assert list(d.keys()) == g_cell["odd_ids"].tolist()
    assert d[77][0] == g_cell["odd_first_rows"].tolist()
    assert d[77][1] == g_cell["odd_first_cols"].tolist()
    loop = rp.cell_positions_loop(g_cell["odd_mask"])
    assert list(loop.keys()) == list(d.keys())
    for k in d:
        assert d[k] == loop[k]


def test_cell_table_empty():
    ids, tab = rp.cell_table(np.zeros((5, 7), np.int32))
    assert ids.shape == (0,) and tab.shape == (0, 7)
    assert rp.cell_positions(np.zeros((5, 7), np.int32)) == {}


@pytest.fixture(scope="module")
def g_patch(golden_dir):
    return np.load(os.path.join(golden_dir, "patches.npz"))


@pytest.mark.parametrize("name", ["all7", "perm", "one_missing", "two_missing", "three"])
def test_patches_match_reference(g_patch, name):
    ids, tab = rp.cell_table(g_patch["A_mask"])
    patches, inten = rp.patches_for_panel(g_patch["A_image"], g_patch["A_mask"], g_patch[f"A_{name}_index"].tolist(), ids, tab)
    np.testing.assert_array_equal(patches, g_patch[f"A_{name}_patches"])
    np.testing.assert_array_equal(inten, g_patch[f"A_{name}_intensity"])


def test_second_missing_marker_aliases_last_channel(g_patch):
    p = g_patch["A_two_missing_patches"]
    full = g_patch["A_all7_patches"]
    assert (p[:, 1] == -1).all()                      # first -1 -> blank plane
    np.testing.assert_array_equal(p[:, 4], full[:, 6])  # second -1 -> last image channel (reference quirk)


def test_patches_unnormalised_uint16(g_patch):
    ids, tab = rp.cell_table(g_patch["B_mask"])
    patches, inten = rp.patches_for_panel(g_patch["B_raw"], g_patch["B_mask"], [2, 0, 1], ids, tab)
    np.testing.assert_array_equal(patches, g_patch["B_patches"])
    np.testing.assert_array_equal(inten, g_patch["B_intensity"])


def test_soft_mask(g_patch):
    lab = np.zeros((40, 40))
    lab[:, :] = g_patch["A_mask"][20:60, 30:70]
    for cid, key in ((65000, "A_smooth_big"), (41, "A_smooth_nested")):
        s = rp.soft_mask(lab, cid)
        assert s.dtype == np.float32
        np.testing.assert_array_equal(s, g_patch[key])
        assert 0.999 < s.max() < 1.0


def test_window_bounds_edges():
    # top/left clamp shifts the window, bottom/right clamp truncates it (utils.py:227-235)
    assert rp.window_bounds(0, 4, 0, 6, 90, 120) == (0, 40, 0, 40)
    assert rp.window_bounds(85, 89, 110, 119, 90, 120) == (67, 90, 94, 120)
    assert rp.window_bounds(30, 39, 40, 49, 90, 120) == (14, 54, 24, 64)


@pytest.mark.parametrize("cell_size", [20, 34, 45, 60])
def test_scaled_patches_match_reference(golden_dir, cell_size):
    """cell_size != 30 through the reference's own _img2patches (preprocess.py:76-135): window int(40 * cell_size / 30), resize to 40."""
    g = np.load(os.path.join(golden_dir, "patches_scaled.npz"))
    ids, tab = rp.cell_table(g["mask"])
    patches, inten = rp.patches_for_panel(g["image"], g["mask"], [2, 0, 1], ids, tab, scale=cell_size / 30.0)
    np.testing.assert_array_equal(patches, g[f"s{cell_size}_patches"])
    np.testing.assert_array_equal(inten, g[f"s{cell_size}_intensity"])
