"""GPU tests of the MX form of the split-operand GEMM (csrc/gemm_mx.hip) through the C ABI: the operand packers bit for bit against
tests/mx_emulation.py, the product against (i) the float64 product of exactly the quantised operands (what the kernel is meant to
compute: only fp32 summation order differs) and (ii) the exact product, with the scheme's error derived at the assert."""
import numpy as np
import pytest
import torch

import mx_emulation as mx
from test_gpu_kernels import _fold, _ln_case, _row_stats, _stats_err, note_err, ps_decode, ps_encode, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from multiplexed_image_annotator_amd import _lib
    return _lib.require_gpu()


def _ps_halves(buf, k):
    """packed-split [R, 2 Kp] int16 -> (hi, lo) float16 arrays [R, k] on the host"""
    r = buf.shape[0]
    f = buf.view(torch.float16).view(r, -1, 2, 8)
    return f[:, :, 0, :].reshape(r, -1)[:, :k].cpu().numpy(), f[:, :, 1, :].reshape(r, -1)[:, :k].cpu().numpy()


def _planes(m, kp128, dev):
    return (torch.zeros((m, kp128), dtype=torch.int16, device=dev), torch.zeros((m, kp128), dtype=torch.uint8, device=dev),
            torch.zeros((m, kp128 // 32), dtype=torch.uint8, device=dev))


def _scales(sc, m):
    """the device's transposed scale plane [Kp / 128][M][4] -> [M, Kp / 32] (numpy uint8)"""
    return sc.view(-1, m, 4).permute(1, 0, 2).reshape(m, -1).cpu().numpy()


@pytest.mark.parametrize("m,k", [(37, 96), (130, 288), (257, 1152), (64, 2304)])
def test_mx_pack_act(dev, m, k):
    """packed-split rows -> MX3 planes: permuted hi plane, e4m3 lo bytes, E8M0 scale bytes, all bit for bit; rows with tiny, huge and
    all-zero blocks included"""
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    kp = (k + 31) // 32 * 32
    kp128 = (k + 127) // 128 * 128
    x = rnd((m, k), 71, dev) * torch.exp(rnd((m, 1), 72, dev) * 3.0)
    x[3, :40] = 0.0
    x[5] *= 1e-6                       # hi subnormal / zero in fp16: lo carries everything that is left
    x[7, 10] = 7.0e4                   # beyond the fp16 range: saturates at 65504 with lo = 0
    a_ps = ps_encode(x, kp)
    hi_p, l8_p, sc_p = _planes(m, kp128, dev)
    check(lib().ribca_test_mx_pack_act(ptr(a_ps), 2 * kp, m, kp, ptr(hi_p), ptr(l8_p), ptr(sc_p), stream_ptr()), "mx_pack_act")
    hi16, lo16 = _ps_halves(a_ps, kp)
    hi_pad = np.zeros((m, kp128), np.float16); hi_pad[:, :kp] = hi16
    lo_pad = np.zeros((m, kp128), np.float64); lo_pad[:, :kp] = lo16.astype(np.float64)
    plane, _, q, sl = mx.pack_act(hi_pad, lo_pad)
    assert np.array_equal(hi_p.cpu().numpy().view(np.float16).view(np.uint16), plane.view(np.uint16))
    assert np.array_equal(_scales(sc_p, m), sl)
    got = mx.e4m3_decode(l8_p.cpu().numpy())
    assert np.array_equal(got, q), np.abs(got - q).max()
    assert np.abs(q).max() <= 256.0      # the scale rule keeps lo / scale away from the e4m3 maximum (the conversion does not saturate)


MX_SHAPES = [(1, 576, 2304), (130, 576, 2304), (300, 384, 1536), (257, 288, 1152), (128, 192, 128), (515, 576, 640), (4000, 576, 2304), (9001, 384, 1536)]


@pytest.mark.parametrize("m,n,k", MX_SHAPES)
@pytest.mark.parametrize("recentre", [False, True])
def test_gemm_mx_resid(dev, m, n, k, recentre):
    """fc2 of the classifiers on the MX kernel: z (packed-split) = (z - previous mean) + A W^T + b, statistics of the new rows"""
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    a = rnd((m, k), 5, dev) * torch.exp(rnd((m, 1), 55, dev))          # rows of different scale
    a = torch.where(rnd((m, k), 56, dev) > 0.5, a * 8.0, a)            # a heavy tail inside the rows, as a GELU output has
    w = rnd((n, k), 6, dev, 1.0 / np.sqrt(k))
    bias = rnd((n,), 7, dev, 0.1)
    npd = (n + 31) // 32 * 32
    z0 = rnd((m, n), 8, dev) + (3.0 if recentre else 0.0)
    a_ps = ps_encode(a, k)
    w_ps = ps_encode(w, k, (n + 15) // 16 * 16)
    z_ps = ps_encode(z0, npd)
    z0q = ps_decode(z_ps, n)
    hi_p, l8_p, sc_p = _planes(m, k, dev)
    wh = torch.zeros(lib().ribca_test_mx_weight_bytes(n, k, 0), dtype=torch.uint8, device=dev)
    wx = torch.zeros(lib().ribca_test_mx_weight_bytes(n, k, 1), dtype=torch.uint8, device=dev)
    tiles = n // 48
    part = torch.zeros((tiles, m, 2), dtype=torch.float32, device=dev)
    rs = torch.zeros((m, 2), dtype=torch.float32, device=dev)
    prev = _row_stats(z_ps, npd, m, n, dev) if recentre else None
    check(lib().ribca_test_gemm_mx_resid(ptr(a_ps), 2 * k, ptr(w_ps), 2 * k, m, n, k, ptr(bias), ptr(hi_p), ptr(l8_p), ptr(sc_p), ptr(wh), ptr(wx),
                                         ptr(z_ps), 2 * npd, ptr(part), ptr(rs), ptr(prev) if recentre else None, stream_ptr()), "gemm_mx_resid")
    got = ps_decode(z_ps, n)
    # (i) the product of exactly the operands the kernel multiplies
    a_hi, a_lo = _ps_halves(a_ps, k)
    w_hi, w_lo = _ps_halves(w_ps, k)
    _, a_lo_q, _, _ = mx.pack_act(a_hi, a_lo.astype(np.float64))
    emu = torch.from_numpy(mx.gemm(a_hi, a_lo_q, w_hi[:n], w_lo[:n])).to(dev)
    shift = prev[:, 1:2].double() if recentre else 0.0
    ref_emu = z0q + emu + bias.double() - shift
    scale = 1.0 + (a.double().abs() @ w.double().abs().t())          # sum_k |a w| bounds every partial sum
    e_emu = ((got - ref_emu).abs() / scale).max().item()
    note_err(f"gemm_mx vs emulated operands {m}x{n}x{k}", e_emu)
    # fp32 accumulation of K / 32 + 2 K / 128 MFMA partial sums + the re-split of the new row (2^-23): a few 1e-7 of sum |a w|
    assert e_emu < 2e-6, e_emu
    # (ii) the exact product: each correction operand is rounded to 4 significant bits (relative 2^-4 at most) and multiplies a factor
    # 2^-11 of the product or less: |error| <= sum_k |a w| * 2^-11 * 2 * 2^-4 * 2 terms = 2^-13 sum |a w| in the worst case, random signs
    # in practice (measured 2-4e-6 of sum |a w|)
    ref = z0q + a.double() @ w.double().t() + bias.double() - shift
    e_ref = ((got - ref).abs() / scale).max().item()
    note_err(f"gemm_mx vs exact product {m}x{n}x{k}", e_ref)
    assert e_ref < 3e-5, e_ref
    assert torch.all(ps_decode(z_ps, npd)[:, n:] == 0)
    e1, e2 = _stats_err(rs, got)
    note_err(f"gemm_mx stats {m}x{n}x{k}", max(e1, e2))
    assert e1 < 1e-6 and e2 < 1e-6, (e1, e2)


@pytest.mark.parametrize("m,d", [(150, 288), (260, 576), (7001, 288), (5000, 576), (12000, 384)])
@pytest.mark.parametrize("mean,std", [(0.5, 3.0), (30.0, 1.0)])
def test_gemm_gelu_mx(dev, m, d, mean, std):
    """norm2 -> mlp.fc1 folded, GELU output straight in the MX3 format: the hi plane and the scale bytes are exactly what the packer makes
    of the packed-split kernel's output (same accumulation order: same x), the lo bytes differ from it only where rounding x - hi to fp16
    first (the packed-split detour) crosses an e4m3 rounding boundary -- x - hi has up to 13 significant bits, fp16 keeps 11, an e4m3 code
    boundary sits every 2^-4 relative: about one value in 64 (measured 1.54-1.60 %), one code step each"""
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    n = 4 * d
    z_ps, zq, g, b, dp = _ln_case(m, d, 50, dev, mean, std, row_scale=(mean == 0.5))
    w = rnd((n, d), 53, dev, 2.0 / np.sqrt(d))
    bias = rnd((n,), 54, dev, 0.1)
    w_ps, csum, bias2 = _fold(w, g, b, bias, dp, dev)
    rs = _row_stats(z_ps, dp, m, d, dev)
    wf = torch.zeros_like(w_ps)
    hi_p, l8_p, sc_p = _planes(m, n, dev)
    check(lib().ribca_test_gemm_gelu_mx(ptr(z_ps), 2 * dp, ptr(w_ps), 2 * dp, m, n, dp, ptr(bias2), ptr(csum), ptr(rs), ptr(wf), ptr(hi_p), ptr(l8_p),
                                        ptr(sc_p), stream_ptr()), "gelu_mx")
    # the packed-split kernel's output of the same product, through the reference packer
    out = torch.zeros((m, 2 * n), dtype=torch.int16, device=dev)
    check(lib().ribca_test_gemm_fold(1, ptr(z_ps), 2 * dp, ptr(w_ps), 2 * dp, m, n, dp, ptr(bias2), ptr(csum), ptr(rs), ptr(out), 2 * n,
                                     stream_ptr()), "gemm_fold")
    hi_r, l8_r, sc_r = _planes(m, n, dev)
    check(lib().ribca_test_mx_pack_act(ptr(out), 2 * n, m, n, ptr(hi_r), ptr(l8_r), ptr(sc_r), stream_ptr()), "mx_pack_act")
    assert torch.equal(hi_p, hi_r)
    assert torch.equal(sc_p, sc_r)
    got, want = mx.e4m3_decode(l8_p.cpu().numpy()), mx.e4m3_decode(l8_r.cpu().numpy())
    diff = got != want
    frac = diff.mean()
    note_err(f"gelu_mx lo codes differing from the packed-split detour {m}x{n}", frac)
    assert frac < 0.03, frac
    # where they differ: by one e4m3 step at that magnitude (subnormal step 2^-9 below 2^-6), plus -- for lo below the fp16 normal
    # range, i.e. |x| < 0.25 -- the 2^-24 quantum of the detour's fp16 lo, which the fused epilogue (fp32 lo) does not have
    scale_e = np.repeat(2.0 ** (_scales(sc_p, m).astype(np.float64) - 127), 32, axis=1)
    step = np.maximum(np.abs(want), 2.0 ** -6) * 2.0 ** -3
    assert np.all((np.abs(got - want) * scale_e)[diff] <= (step * scale_e)[diff] * 1.01 + 2.0 ** -24)
    # and the operand as fc2 will see it against the exact product
    ln = torch.nn.functional.layer_norm(zq, (d,), g.double(), b.double(), 1e-6)
    ref = torch.nn.functional.gelu(ln @ w.double().t() + bias.double()).cpu().numpy()
    hi = hi_p.cpu().numpy().view(np.float16)[:, mx.hi_pos(np.arange(n))].astype(np.float64)
    scale = 2.0 ** (_scales(sc_p, m).astype(np.float64) - 127)
    val = hi + (got.reshape(m, n // 32, 32) * scale[:, :, None]).reshape(m, n)
    err = np.abs(val - ref).max()
    note_err(f"gelu_mx hi + lo vs exact {m}x{n}x{d} mean {mean}", err)
    # test_gemm_fold_gelu's bound plus the e4m3 rounding of lo: 2^-4 of |lo| <= 2^-16 of the block's largest value
    assert err < 2e-5 * (1.0 + abs(mean) / std) + 2.0 ** -15 * np.abs(ref).max(), err
