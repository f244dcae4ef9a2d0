"""GPU tests of the MX form of the split-operand GEMM (csrc/gemm_mx.hip) through the C ABI: the operand packers bit for bit against
tests/mx_emulation.py, the product against (i) the float64 product of exactly the quantised operands (what the kernel is meant to
compute: only fp32 summation order differs) and (ii) the exact product, with the scheme's error derived at the assert."""
import numpy as np
import pytest
import torch

import mx_emulation as mx
from test_gpu_kernels import _row_stats, _stats_err, note_err, ps_decode, ps_encode, rnd

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
    assert np.array_equal(sc_p.cpu().numpy(), sl)
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
