"""GPU tests of the library's robustness contract (include/ribca_hip.h: "no entry point ends the calling process"):

* every operand of the ragged-row cases (M = 1, 101, 127, 129: a single partial 128-row tile, one row short of a tile, one row beyond)
  sits between GUARD BANDS, and every launch runs twice -- bands of zeros, bands of 0x7C bytes (NaN as fp16 halves, 5e36 as fp32,
  a huge e4m3 code): outputs must be bit-identical (no read from beyond an operand reaches a result) and the bands untouched (no write
  beyond an output).  This is the deterministic form of the question round 4's unexplained process abort left open:
  it needs no fault to show an out-of-bounds access, and it shows the ones a caching allocator hides.
* requests the library has no kernel for come back as a status with a message, never as a process abort (the six abort() guards of
  round 4: VERDICT r4 weak #3 / ADVICE r4).
"""
import numpy as np
import pytest
import torch

from test_gpu_kernels import _fold, _ln_case, _row_stats, ps_encode, rnd

pytestmark = pytest.mark.gpu

BAND = 8192      # bytes on either side of every operand: more than a 128-row tile of 48-byte rows, a 4 KB page and a 1 KB DMA piece


@pytest.fixture(scope="module")
def dev():
    from multiplexed_image_annotator_amd import _lib
    return _lib.require_gpu()


class Arena:
    """operands carved out of ONE device buffer with a band of `fill` bytes before and after each"""

    def __init__(self, dev, fill, capacity=96 << 20):
        self.buf = torch.full((capacity,), fill, dtype=torch.uint8, device=dev)
        self.fill, self.off, self.spans = fill, BAND, []

    def put(self, t: torch.Tensor) -> torch.Tensor:
        """a copy of t inside the arena"""
        v = self.empty(t.shape, t.dtype)
        v.copy_(t)
        return v

    def zeros(self, shape, dtype):
        v = self.empty(shape, dtype)
        v.zero_()
        return v

    def empty(self, shape, dtype):
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        start = (self.off + 255) // 256 * 256
        assert start + nbytes + BAND <= self.buf.numel(), "arena too small"
        self.spans.append((start, nbytes))
        self.off = start + nbytes + BAND
        return self.buf[start:start + nbytes].view(dtype).view(shape)

    def bands_intact(self):
        """every byte outside the carved spans still holds the fill pattern"""
        mask = torch.ones(self.off, dtype=torch.bool, device=self.buf.device)
        for s, n in self.spans:
            mask[s:s + n] = False
        return bool(torch.all(self.buf[:self.off][mask] == self.fill))


def _twice(dev, body):
    """body(arena) -> dict of output tensors; run with zero bands and with 0x7C bands: same bits, bands intact"""
    outs = []
    for fill in (0x00, 0x7C):
        ar = Arena(dev, fill)
        res = body(ar)
        torch.cuda.synchronize()
        assert ar.bands_intact(), f"a kernel wrote outside its operands (band fill {fill:#x})"
        outs.append({k: v.clone() for k, v in res.items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), f"{k} depends on bytes beyond an operand"
    return outs[0]


def _planes(ar, m, kp128):
    return ar.zeros((m, kp128), torch.int16), ar.zeros((m, kp128), torch.uint8), ar.zeros((m, kp128 // 32), torch.uint8)


@pytest.mark.parametrize("d", [384, 576])
def test_qkv_attention_mx_one_cell_between_guard_bands(dev, d):
    """the case that ended the process once in round 4 (gpurun_out/mx_tests.log, d = 384, one cell: M = 101 rows, one ragged tile)"""
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    heads, ntok, cells = 12, 101, 1
    hdp = (d // heads + 7) // 8 * 8
    m = cells * ntok
    z_ps, _, g, b, dp = _ln_case(m, d, 60, dev, 0.5, 1.0)
    w = rnd((3 * d, d), 63, dev, 1.0 / np.sqrt(d))
    bias = rnd((3 * d,), 64, dev, 0.1)
    w_ps, csum, bias2 = _fold(w, g, b, bias, dp, dev)
    rs = _row_stats(z_ps, dp, m, d, dev)
    kz = (dp + 127) // 128 * 128

    def body(ar):
        z, wp, cs, b2, r = ar.put(z_ps), ar.put(w_ps), ar.put(csum), ar.put(bias2), ar.put(rs)
        a_hi, a_l8, a_sc = _planes(ar, m, kz)
        wh = ar.zeros((lib().ribca_test_mx_weight_bytes(3 * d, kz, 0),), torch.uint8)
        wx = ar.zeros((lib().ribca_test_mx_weight_bytes(3 * d, kz, 1),), torch.uint8)
        q, k, vt = (ar.zeros((cells, heads, 112, 2 * hdp), torch.int16) for _ in range(3))
        out = ar.zeros((m, 2 * dp), torch.int16)
        check(lib().ribca_test_qkv_attention_mx(ptr(z), 2 * dp, ptr(wp), 2 * dp, cells, d, dp, ptr(b2), ptr(cs), ptr(r), ptr(a_hi), ptr(a_l8), ptr(a_sc),
                                                ptr(wh), ptr(wx), ptr(q), ptr(k), ptr(vt), ptr(out), 2 * dp, stream_ptr()), "qkv mx")
        return {"q": q, "k": k, "v": vt, "out": out, "a_hi": a_hi, "a_l8": a_l8, "a_sc": a_sc}

    res = _twice(dev, body)
    assert torch.any(res["out"] != 0)


@pytest.mark.parametrize("m", [1, 101, 127, 129])
def test_gemm_mx_fc1_ragged_rows_between_guard_bands(dev, m):
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    d = 384
    n = 4 * d
    z_ps, _, g, b, dp = _ln_case(m, d, 50, dev, 0.5, 3.0)
    w = rnd((n, d), 53, dev, 2.0 / np.sqrt(d))
    bias = rnd((n,), 54, dev, 0.1)
    w_ps, csum, bias2 = _fold(w, g, b, bias, dp, dev)
    rs = _row_stats(z_ps, dp, m, d, dev)
    kz = (dp + 127) // 128 * 128

    def body(ar):
        z, wp, cs, b2, r = ar.put(z_ps), ar.put(w_ps), ar.put(csum), ar.put(bias2), ar.put(rs)
        a_hi, a_l8, a_sc = _planes(ar, m, kz)
        wh = ar.zeros((lib().ribca_test_mx_weight_bytes(n, kz, 0),), torch.uint8)
        wx = ar.zeros((lib().ribca_test_mx_weight_bytes(n, kz, 1),), torch.uint8)
        hi_p, l8_p, sc_p = _planes(ar, m, n)
        check(lib().ribca_test_gemm_mx_fc1(ptr(z), 2 * dp, ptr(wp), 2 * dp, m, n, dp, ptr(b2), ptr(cs), ptr(r), ptr(a_hi), ptr(a_l8), ptr(a_sc), ptr(wh),
                                           ptr(wx), ptr(hi_p), ptr(l8_p), ptr(sc_p), stream_ptr()), "gemm_mx_fc1")
        return {"hi": hi_p, "l8": l8_p, "sc": sc_p}

    res = _twice(dev, body)
    assert torch.any(res["hi"] != 0)


@pytest.mark.parametrize("kind,n,k", [(0, 576, 576), (1, 576, 2304), (1, 384, 1536)])
@pytest.mark.parametrize("m", [1, 101, 127, 129])
def test_gemm_resid_zmx_ragged_rows_between_guard_bands(dev, kind, n, k, m):
    """attn.proj (two-workgroups kernel) / mlp.fc2 (MX kernel) with the residual tile through the ring, statistics and the MX3 copy"""
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    a = rnd((m, k), 5, dev) * torch.exp(rnd((m, 1), 55, dev))
    w = rnd((n, k), 6, dev, 1.0 / np.sqrt(k))
    bias = rnd((n,), 7, dev, 0.1)
    npd = (n + 31) // 32 * 32
    zk = (n + 127) // 128 * 128
    z0 = (rnd((m, n), 8, dev) + 3.0) * torch.exp(rnd((m, 1), 58, dev) * 2.0)
    a_ps, w_ps, z_ps = ps_encode(a, k), ps_encode(w, k, lib().ribca_gemm_padded_n(n)), ps_encode(z0, npd)
    prev = _row_stats(z_ps, npd, m, n, dev)

    def body(ar):
        ap, wp, zp, bs, pv = ar.put(a_ps), ar.put(w_ps), ar.put(z_ps), ar.put(bias), ar.put(prev)
        part = ar.zeros((n // 48, m, 2), torch.float32)
        rs = ar.zeros((m, 2), torch.float32)
        z_hi, z_l8, z_sc = _planes(ar, m, zk)
        a_hi, a_l8, a_sc = _planes(ar, m, k if k % 128 == 0 else 128)
        if kind == 0:
            wsc, wx = ar.zeros(tuple(w_ps.shape), torch.int16), ar.zeros((16,), torch.uint8)
        else:
            wsc = ar.zeros((lib().ribca_test_mx_weight_bytes(n, k, 0),), torch.uint8)
            wx = ar.zeros((lib().ribca_test_mx_weight_bytes(n, k, 1),), torch.uint8)
        check(lib().ribca_test_gemm_resid_zmx(kind, ptr(ap), 2 * k, ptr(wp), 2 * k, m, n, k, ptr(bs), ptr(a_hi), ptr(a_l8), ptr(a_sc), ptr(wsc), ptr(wx),
                                              ptr(zp), 2 * npd, ptr(part), ptr(rs), ptr(pv), ptr(z_hi), ptr(z_l8), ptr(z_sc), zk, stream_ptr()),
              "gemm_resid_zmx")
        return {"z": zp, "rs": rs, "part": part, "z_hi": z_hi, "z_l8": z_l8, "z_sc": z_sc}

    res = _twice(dev, body)
    assert torch.any(res["z"] != z_ps)


@pytest.mark.parametrize("d", [144, 288, 384])
def test_cell_attention_one_cell_between_guard_bands(dev, d):
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    m = 101
    z_ps, _, g, b, dp = _ln_case(m, d, 70, dev, 0.5, 1.0)
    w = rnd((3 * d, d), 73, dev, 1.0 / np.sqrt(d))
    bias = rnd((3 * d,), 74, dev, 0.1)
    w_ps, csum, bias2 = _fold(w, g, b, bias, dp, dev)
    rs = _row_stats(z_ps, dp, m, d, dev)

    def body(ar):
        z, wp, cs, b2, r = ar.put(z_ps), ar.put(w_ps), ar.put(csum), ar.put(bias2), ar.put(rs)
        out = ar.zeros((m, 2 * dp), torch.int16)
        check(lib().ribca_test_cell_attention(ptr(z), 2 * dp, ptr(wp), 2 * dp, 1, d, ptr(b2), ptr(cs), ptr(r), ptr(out), 2 * dp, stream_ptr()), "cell attention")
        return {"out": out}

    res = _twice(dev, body)
    assert torch.any(res["out"] != 0)


# ------------------------------------------------------------------------------------------------ refused requests are statuses
def test_gemm_variant_on_an_mx_model_is_a_status_not_an_abort(dev):
    """ribca_set_gemm_variant(v != 0) followed by ribca_vit_forward on a D = 384 model reached `g_variant != 0 -> abort()` in
    launch_gemm_resid_ps (the MX3 copy of the residual rows exists on the production form only) and killed the interpreter"""
    from multiplexed_image_annotator_amd import ops, synth
    from multiplexed_image_annotator_amd._lib import RibcaError, lib
    if not lib().ribca_mxz_enabled(384):
        pytest.skip("RIBCA_MX / RIBCA_MXZ switched off: the residual rows are not kept in MX3")
    sd = synth.make_vit_state_dict("immune_extended", synth.SEED_BASE + 3, depth=3)
    model = ops.VitModel(sd, dev)
    patches = rnd((5, model.C, 40, 40), 91, dev)
    good = model.predict_proba(patches, list(range(model.C)))
    lib().ribca_set_gemm_variant(3)
    try:
        with pytest.raises(RibcaError, match="128 x 192"):
            model.predict_proba(patches, list(range(model.C)))
        torch.cuda.synchronize()
    finally:
        lib().ribca_set_gemm_variant(0)
    again = model.predict_proba(patches, list(range(model.C)))
    assert torch.equal(good, again)      # the error was consumed: nothing stale is reported by the next call


def test_unsupported_attention_geometry_is_a_status(dev):
    """launch_attention used to abort() on a head dimension it has no kernel for (D = 96: hd = 8)"""
    from multiplexed_image_annotator_amd._lib import lib, ptr, stream_ptr
    d, cells = 96, 1
    m = cells * 101
    a_ps = ps_encode(rnd((m, d), 1, dev), d)
    w_ps = ps_encode(rnd((3 * d, d), 2, dev, 0.1), d, lib().ribca_gemm_padded_n(3 * d))
    bias = rnd((3 * d,), 3, dev, 0.1)
    q, k, vt = (torch.zeros((cells, 12, 112, 2 * 32), dtype=torch.int16, device=dev) for _ in range(3))      # (generous: hd 8 needs 16 per row)
    out = torch.zeros((m, 2 * d), dtype=torch.int16, device=dev)
    st = lib().ribca_test_qkv_attention(ptr(a_ps), 2 * d, ptr(w_ps), 2 * d, cells, d, d, ptr(bias), ptr(q), ptr(k), ptr(vt), ptr(out), 2 * d, stream_ptr())
    torch.cuda.synchronize()
    assert st != 0 and b"launch_attention" in lib().ribca_last_error()
    assert torch.all(out == 0)


def test_mx_launchers_refuse_shapes_without_a_tile_form(dev):
    """N % 192 != 0 on the MX forms of qkv / fc1 / the MX3 copy: status + message (round 4: fprintf + abort())"""
    from multiplexed_image_annotator_amd._lib import lib, ptr, stream_ptr
    m, d = 64, 132      # 4 d = 528 = 11 x 48: a multiple of the MX kernel's wave block, not of its 192-column tile
    n = 4 * d
    z_ps, _, g, b, dp = _ln_case(m, d, 50, dev)
    w = rnd((n, d), 53, dev, 0.1)
    bias = rnd((n,), 54, dev, 0.1)
    w_ps, csum, bias2 = _fold(w, g, b, bias, dp, dev)
    rs = _row_stats(z_ps, dp, m, d, dev)
    kz = (dp + 127) // 128 * 128
    z3 = lambda *s, dt=torch.uint8: torch.zeros(s, dtype=dt, device=dev)
    a_hi, a_l8, a_sc = z3(m, kz, dt=torch.int16), z3(m, kz), z3(m, kz // 32)
    wh, wx = z3(lib().ribca_test_mx_weight_bytes(n, kz, 0)), z3(lib().ribca_test_mx_weight_bytes(n, kz, 1))
    hi_p, l8_p, sc_p = z3(m, n + 192, dt=torch.int16), z3(m, n + 192), z3(m, (n + 192) // 32)
    st = lib().ribca_test_gemm_mx_fc1(ptr(z_ps), 2 * dp, ptr(w_ps), 2 * dp, m, n, dp, ptr(bias2), ptr(csum), ptr(rs), ptr(a_hi), ptr(a_l8), ptr(a_sc), ptr(wh),
                                      ptr(wx), ptr(hi_p), ptr(l8_p), ptr(sc_p), stream_ptr())
    torch.cuda.synchronize()
    assert st != 0 and b"192" in lib().ribca_last_error()
    assert torch.all(hi_p == 0)
