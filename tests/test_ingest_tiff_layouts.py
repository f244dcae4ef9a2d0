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


# ---- what a description announces, and files that cannot be decoded (stacks must not become channels, decode errors must
# reach the fallback reader as TiffUnsupported)
def _ome(c, z=1, t=1):
    return (f'<?xml version="1.0"?><OME xmlns="http://www.openmicroscopy.org/Schemas/OME/2016-06"><Image ID="Image:0"><Pixels ID="Pixels:0" '
            f'DimensionOrder="XYCZT" Type="uint16" SizeX="24" SizeY="20" SizeC="{c}" SizeZ="{z}" SizeT="{t}"></Pixels></Image></OME>')


def test_ome_description_with_z_planes_is_rejected(tmp_path):
    """an OME stack with SizeZ = 2 is 6 pages that are NOT 6 channels: TiffStackError (a ValueError), not a (6, H, W) image"""
    planes = [_rand((20, 24), np.uint16, 100 + i) for i in range(6)]
    path = str(tmp_path / "z.ome.tif")
    write_tiff(path, planes, descriptions=[_ome(3, z=2)] + [None] * 5)
    with pytest.raises(pp.TiffStackError):
        pp.read_tiff(path)
    with pytest.raises(ValueError):
        pp.read_image(path)                       # and read_image does not hand the file to a reader that would flatten it
    path_t = str(tmp_path / "t.ome.tif")
    write_tiff(path_t, planes, descriptions=[_ome(2, t=3)] + [None] * 5)
    with pytest.raises(pp.TiffStackError):
        pp.read_tiff(path_t)


def test_ome_channel_count_must_match_the_pages(tmp_path):
    planes = [_rand((20, 24), np.uint16, 110 + i) for i in range(4)]
    ok = str(tmp_path / "c4.ome.tif")
    write_tiff(ok, planes, descriptions=[_ome(4)] + [None] * 3)
    np.testing.assert_array_equal(pp.read_tiff(ok), np.stack(planes))
    bad = str(tmp_path / "c5.ome.tif")
    write_tiff(bad, planes, descriptions=[_ome(5)] + [None] * 3)
    with pytest.raises(pp.TiffStackError) as e:   # not TiffUnsupported: another decoder reading page by page would return a partial image
        pp.read_tiff(bad)
    assert "4 page(s) of shape (20, 24)" in str(e.value) and "companion" in str(e.value)      # the message says what WAS found
    with pytest.raises(ValueError):
        pp.read_image(bad)
    # MORE same-shape pages than SizeC: the first series is the first SizeC of them, as tifffile returns it
    extra = str(tmp_path / "c3_of_4.ome.tif")
    write_tiff(extra, planes, descriptions=[_ome(3)] + [None] * 3)
    np.testing.assert_array_equal(pp.read_tiff(extra), np.stack(planes[:3]))


def _set_compression(path, code):
    """patch the Compression tag (259) of every IFD of a classic little-endian file written by write_tiff"""
    raw = bytearray(open(path, "rb").read())
    ifd = struct.unpack("<I", raw[4:8])[0]
    while ifd:
        n = struct.unpack("<H", raw[ifd:ifd + 2])[0]
        for i in range(n):
            e = ifd + 2 + 12 * i
            if struct.unpack("<H", raw[e:e + 2])[0] == 259:
                raw[e + 8:e + 10] = struct.pack("<H", code)
        ifd = struct.unpack("<I", raw[ifd + 2 + 12 * n:ifd + 6 + 12 * n])[0]
    open(path, "wb").write(bytes(raw))


def test_z_stack_in_an_undecoded_compression_is_still_a_stack(tmp_path):
    """the Z / T check ran only after every page's layout had been accepted, so an LZW- or JPEG-compressed OME Z stack raised
    TiffUnsupported and read_image() handed it to PIL, which stacks the Z planes as channels"""
    planes = [_rand((20, 24), np.uint16, 140 + i) for i in range(3)]
    path = str(tmp_path / "z_lzw.ome.tif")
    write_tiff(path, planes, descriptions=[_ome(1, z=3)] + [None] * 2)
    _set_compression(path, 5)                     # LZW: this reader does not decode it
    with pytest.raises(pp.TiffStackError):
        pp.read_tiff(path)
    with pytest.raises(ValueError):
        pp.read_image(path)
    plain = str(tmp_path / "c3_lzw.ome.tif")      # the same compression on a plain (C, H, W) file is merely unsupported (fallback allowed)
    write_tiff(plain, planes, descriptions=[_ome(3)] + [None] * 2)
    _set_compression(plain, 5)
    with pytest.raises(pp.TiffUnsupported):
        pp.read_tiff(plain)


def test_imagej_stack_behind_one_ifd(tmp_path):
    """ImageJ writes large stacks as ONE IFD followed by images=N contiguous planes: (N, H, W) like tifffile, not the first plane alone"""
    planes = [_rand((20, 24), np.uint16, 120 + i) for i in range(5)]
    path = str(tmp_path / "ij.tif")
    # one page whose single strip covers the image, then the other planes appended right behind the file (the writer puts the IFD after
    # the pixel data, so the stack is rebuilt here: header + all planes + IFD of plane 0)
    write_tiff(path, [planes[0]], descriptions=["ImageJ=1.53t\nimages=5\nchannels=5\nhyperstack=true\n"])
    with pytest.raises(pp.TiffUnsupported):
        pp.read_tiff(path)                                 # announces 5 images, holds 1: does not fit
    # a real contiguous stack: strips of plane 0 are 16 rows each and contiguous from offset 8; put the other planes where the writer
    # put the IFD by writing a file whose "first page" is the whole stack as one tall image, then patching ImageLength
    tall = np.concatenate(planes, axis=0)
    write_tiff(path, [tall], descriptions=["ImageJ=1.53t\nimages=5\nchannels=5\nhyperstack=true\n"])
    raw = bytearray(open(path, "rb").read())
    ifd = struct.unpack("<I", raw[4:8])[0]
    n = struct.unpack("<H", raw[ifd:ifd + 2])[0]
    for i in range(n):
        e = ifd + 2 + 12 * i
        tag = struct.unpack("<H", raw[e:e + 2])[0]
        if tag == 257:                                     # ImageLength: one plane
            raw[e + 8:e + 12] = struct.pack("<I", 20)
        if tag == 278:                                     # RowsPerStrip: the plane in one strip
            raw[e + 8:e + 12] = struct.pack("<I", 20)
        if tag in (273, 279):                              # one strip: first offset / one plane of bytes, stored in line
            cnt, typ = struct.unpack("<I", raw[e + 4:e + 8])[0], struct.unpack("<H", raw[e + 2:e + 4])[0]
            first = struct.unpack("<I", raw[struct.unpack("<I", raw[e + 8:e + 12])[0]:][:4])[0] if cnt > 1 else struct.unpack("<I", raw[e + 8:e + 12])[0]
            raw[e + 4:e + 8] = struct.pack("<I", 1)
            raw[e + 8:e + 12] = struct.pack("<I", first if tag == 273 else 20 * 24 * 2)
    open(path, "wb").write(bytes(raw))
    got = pp.read_tiff(path)
    assert got.shape == (5, 20, 24)
    np.testing.assert_array_equal(got, np.stack(planes))


def test_imagej_hyperstack_with_slices_is_rejected(tmp_path):
    planes = [_rand((20, 24), np.uint16, 130 + i) for i in range(6)]
    path = str(tmp_path / "ij_z.tif")
    write_tiff(path, planes, descriptions=["ImageJ=1.53t\nimages=6\nchannels=2\nslices=3\nhyperstack=true\n"] + [None] * 5)
    with pytest.raises(pp.TiffStackError):
        pp.read_tiff(path)


@pytest.mark.parametrize("deflate", [False, True])
def test_truncated_and_corrupt_files_are_declined_not_crashed(tmp_path, deflate):
    """struct / zlib / key / buffer-size errors of a damaged file surface as TiffUnsupported (so that read_image can try the next reader),
    never as struct.error, zlib.error, KeyError or a numpy buffer ValueError"""
    planes = [_rand((40, 24), np.uint16, 140 + i) for i in range(3)]
    path = str(tmp_path / "whole.tif")
    write_tiff(path, planes, deflate=deflate)
    raw = open(path, "rb").read()
    np.testing.assert_array_equal(pp.read_tiff(path), np.stack(planes))
    ifd = struct.unpack("<I", raw[4:8])[0]
    for cut in (6, 100, ifd + 1, len(raw) // 2, len(raw) - 3):
        part = str(tmp_path / f"cut{cut}.tif")
        open(part, "wb").write(raw[:cut])
        try:
            got = pp.read_tiff(part)                       # a cut behind the last byte that matters may still decode ...
            assert got.shape[-2:] == (40, 24)
        except pp.TiffUnsupported:
            pass                                           # ... everything else is declined with the one exception type
    # a strip pointing into garbage
    bad = bytearray(raw)
    bad[8:40] = b"\xff" * 32
    p2 = str(tmp_path / "garbage.tif")
    open(p2, "wb").write(bytes(bad))
    if deflate:
        with pytest.raises(pp.TiffUnsupported):
            pp.read_tiff(p2)
    # StripOffsets removed (tag renamed to a private one)
    miss = bytearray(raw)
    n = struct.unpack("<H", miss[ifd:ifd + 2])[0]
    for i in range(n):
        e = ifd + 2 + 12 * i
        if struct.unpack("<H", miss[e:e + 2])[0] == 273:
            miss[e:e + 2] = struct.pack("<H", 65000)
    p3 = str(tmp_path / "nooffsets.tif")
    open(p3, "wb").write(bytes(miss))
    with pytest.raises(pp.TiffUnsupported):
        pp.read_tiff(p3)
