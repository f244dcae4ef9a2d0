"""Image / mask ingest of the drop-in (reference preprocess.py:244-250: ``imread`` of a multi-channel TIFF, ``(H, W, 3)`` mask ->
first channel, ``int32`` cast): files written here with PIL in the layouts the reference's users have -- multi-page uint16 TIFF,
16-bit PNG label mask, RGB PNG mask, float32 TIFF -- must come back as the arrays that were written, in page order."""
import numpy as np
import pytest

from multiplexed_image_annotator_amd import preprocess as pp

PIL = pytest.importorskip("PIL.Image")


def _planes(c, h, w, seed, dtype):
    rng = np.random.default_rng(seed)
    if np.issubdtype(dtype, np.floating):
        return rng.random((c, h, w)).astype(dtype) * 1000
    return rng.integers(0, np.iinfo(dtype).max, (c, h, w), dtype=dtype)


def _save_pages(path, planes):
    pages = [PIL.fromarray(p) for p in planes]
    pages[0].save(path, save_all=True, append_images=pages[1:])


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8, np.float32])
def test_multipage_tiff_is_channel_first_in_page_order(tmp_path, dtype):
    planes = _planes(7, 37, 53, 1, dtype)
    path = str(tmp_path / "img.tif")
    _save_pages(path, planes)
    got = pp.read_image(path)
    assert got.dtype == dtype and got.shape == (7, 37, 53)
    np.testing.assert_array_equal(got, planes)


def test_single_page_tiff_becomes_one_plane(tmp_path):
    plane = _planes(1, 20, 30, 2, np.uint16)[0]
    path = str(tmp_path / "one.tiff")
    PIL.fromarray(plane).save(path)
    got = pp.as_channel_planes(pp.read_image(path), path)
    assert got.shape == (1, 20, 30)
    np.testing.assert_array_equal(got[0], plane)


def test_png_label_mask_16bit_and_rgb(tmp_path):
    rng = np.random.default_rng(3)
    lab = rng.integers(0, 3000, (41, 29)).astype(np.uint16)          # more than 255 cells: needs the 16-bit PNG
    PIL.fromarray(lab).save(str(tmp_path / "m16.png"))
    got = pp.read_image(str(tmp_path / "m16.png"))
    assert got.shape == (41, 29)
    np.testing.assert_array_equal(got.astype(np.int32), lab.astype(np.int32))
    rgb = np.stack([lab.astype(np.uint8), np.zeros_like(lab, np.uint8), np.full_like(lab, 9, np.uint8)], axis=-1)
    PIL.fromarray(rgb).save(str(tmp_path / "mrgb.png"))
    got = pp.read_image(str(tmp_path / "mrgb.png"))
    assert got.shape == (41, 29, 3)
    np.testing.assert_array_equal(got[:, :, 0], rgb[:, :, 0])       # transform() keeps channel 0 (preprocess.py:247-248)


def test_reference_example_mask_layout():
    """Same reader on a label PNG of the kind the reference ships (examples/example_*_cell_mask.png: 16-bit grey): the fixture
    holds the decoded array of example_2, re-encode it and read it back."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cellpos.npz"))
    ex = g["example2_mask"]
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        PIL.fromarray(ex.astype(np.uint16)).save(os.path.join(d, "ex.png"))
        got = pp.read_image(os.path.join(d, "ex.png"))
    np.testing.assert_array_equal(got, ex)


def test_bad_rank_is_rejected(tmp_path):
    np.save(str(tmp_path / "bad.npy"), np.zeros((2, 3, 4, 5), np.uint16))
    with pytest.raises(ValueError):
        pp.as_channel_planes(pp.read_image(str(tmp_path / "bad.npy")), "bad.npy")
