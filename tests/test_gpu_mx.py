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


def _lo_codes_close(l8_p, l8_r, sc_p, m, what):
    """lo bytes of a fused MX3 producer against the packer's on the packed-split form of the same values: equal except where rounding
    x - hi to fp16 first (the packed-split detour) crosses an e4m3 rounding boundary -- about one value in 64 -- and there by one e4m3 step
    at that magnitude (subnormal step 2^-9 below 2^-6), plus -- for lo below the fp16 normal range -- the 2^-24 quantum of the detour's
    fp16 lo, which a fused epilogue (fp32 lo) does not have.  Returns the decoded bytes of the fused producer."""
    got, want = mx.e4m3_decode(l8_p.cpu().numpy()), mx.e4m3_decode(l8_r.cpu().numpy())
    diff = got != want
    scale_e = np.repeat(2.0 ** (_scales(sc_p, m).astype(np.float64) - 127), 32, axis=1)
    # The double-rounding argument (one value in 64) holds where the detour's lo is a NORMAL fp16.  Where it is subnormal (|lo| < 2^-14: the
    # small outputs of a GELU, whose value a.s - 0.5 |x| cancels 8-12 bits) the detour quantises lo to 2^-24 and a fused epilogue does not:
    # since round 6 the GELU's last step is ONE fused multiply-add (ribca_common.h gelu_erf1; the packed form of rounds 2-5 rounded the
    # product first, so its small outputs were multiples of 2^-24 themselves and the two producers agreed by accident of that coarser
    # arithmetic) -- there only the absolute bound below applies.
    normal = np.abs(want * scale_e) >= 2.0 ** -14
    frac = diff[normal].mean() if normal.any() else 0.0
    note_err(f"{what}: lo codes differing from the packed-split detour (normal-fp16 lo; {diff[~normal].mean() if (~normal).any() else 0.0:.3f} of the "
             f"{(~normal).mean():.3f} subnormal-lo values differ)", frac)
    assert frac < 0.03, frac
    step = np.maximum(np.abs(want), 2.0 ** -6) * 2.0 ** -3
    assert np.all((np.abs(got - want) * scale_e)[diff] <= (step * scale_e)[diff] * 1.01 + 2.0 ** -24)
    return got


def _mx3_value(hi_p, l8_p, sc_p, m, n):
    """hi + lo * 2^(scale byte - 127) of the first n columns of an MX3 triple (numpy float64 [m, n])"""
    kp = hi_p.shape[1]
    hi = hi_p.cpu().numpy().view(np.float16)[:, mx.hi_pos(np.arange(kp))].astype(np.float64)
    scale = 2.0 ** (_scales(sc_p, m).astype(np.float64) - 127)
    lo = mx.e4m3_decode(l8_p.cpu().numpy())
    return (hi + (lo.reshape(m, kp // 32, 32) * scale[:, :, None]).reshape(m, kp))[:, :n]


def _scale_rule_holds(hi_p, sc_p, m, n):
    """every scale byte of the first n columns is the format's function of the block's largest |hi|"""
    kp = hi_p.shape[1]
    bits = hi_p.cpu().numpy().view(np.uint16)[:, mx.hi_pos(np.arange(kp))] & 0x7fff
    ef = (bits.reshape(m, kp // 32, 32).max(axis=2) >> 10).astype(np.int64)
    want = np.maximum(ef, 1) + 93
    return np.array_equal(_scales(sc_p, m)[:, : n // 32], want[:, : n // 32].astype(np.uint8))


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


def test_mx_pack_act_outlier_blocks(dev):
    """32-column blocks with ONE element 2^12 above the rest (a massive-activation channel of a trained ViT): the block's
    scale follows the outlier, so the small elements' lo sits 2^12 lower under the shared E8M0 scale -- still inside e4m3's subnormal range
    (lo / scale >= 2^-9), never flushed.  Planes bit for bit against the emulation; the small elements' MX3 value keeps >= 11 + 3 bits."""
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    m, k = 96, 1152
    x = rnd((m, k), 171, dev) * 0.02
    cols = torch.arange(5, k, 32, device=dev)                      # one outlier per 32-column block
    x[:, cols] = x[:, cols] * 4096.0
    x[7, :] = x[7, :] * 1e-3                                        # a row whose small elements are fp16-subnormal in hi as well
    a_ps = ps_encode(x, k)
    hi_p, l8_p, sc_p = _planes(m, k, dev)
    check(lib().ribca_test_mx_pack_act(ptr(a_ps), 2 * k, m, k, ptr(hi_p), ptr(l8_p), ptr(sc_p), stream_ptr()), "mx_pack_act")
    hi16, lo16 = _ps_halves(a_ps, k)
    plane, lo_deq, q, sl = mx.pack_act(hi16.copy(), lo16.astype(np.float64))
    assert np.array_equal(hi_p.cpu().numpy().view(np.float16).view(np.uint16), plane.view(np.uint16))
    assert np.array_equal(_scales(sc_p, m), sl)
    assert np.array_equal(mx.e4m3_decode(l8_p.cpu().numpy()), q)
    # what the format keeps of the SMALL elements of such a block: their lo is rounded on the e4m3 grid of the OUTLIER's scale -- one step of
    # lo (2^-4 relative) where lo / scale is a normal e4m3, half the subnormal quantum (2^-10 scale = 2^-29 of the outlier) where it is not.
    # Measured: the smallest elements (|x| ~ 2^-20 of the outlier) keep hi only (relative 2^-11), a typical small one hi + 5 bits of lo.
    val = hi16.astype(np.float64) + lo_deq
    exact = hi16.astype(np.float64) + lo16.astype(np.float64)
    scale = np.repeat(2.0 ** (sl.astype(np.float64) - 127), 32, axis=1)
    bound = np.maximum(np.abs(lo16.astype(np.float64)) * 2.0 ** -4, scale * 2.0 ** -10)
    assert np.all(np.abs(val - exact) <= bound * 1.0001)
    small = np.ones((m, k), bool); small[:, cols.cpu().numpy()] = False; small[7] = False
    rel = np.abs(val - exact)[small] / np.maximum(np.abs(exact)[small], 1e-30)
    note_err("mx3 small elements beside a 2^12 outlier: relative error of hi + lo' (max; median %.1e)" % float(np.median(rel)), rel.max())
    assert rel.max() <= 2.0 ** -11 * 1.001, rel.max()      # never worse than hi alone


def test_gemm_mx_resid_outlier_operands(dev):
    """the MX product with massive-activation columns in A (4 columns 50 x the rest, as synth.make_vit_state_dict_heavy's residual channels)
    and heavy-tailed (Student-t(3)) weights: against the exact product within the scheme's 2^-15 class of sum |a w| -- the small elements of
    an outlier's block lose their fp6 image (flushed under the block scale), i.e. their two correction terms, which is 2^-11 of THEIR
    products and nothing beside the outlier's own"""
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    m, n, k = 515, 576, 2304
    a = rnd((m, k), 181, dev) * 0.1
    a[:, [37, 700, 1501, 2222]] *= 50.0
    num = rnd((n, k), 182, dev)
    den = torch.sqrt((rnd((n, k), 183, dev) ** 2 + rnd((n, k), 184, dev) ** 2 + rnd((n, k), 185, dev) ** 2) / 3.0).clamp_min(1e-3)
    w = (num / den) / np.sqrt(3.0) / np.sqrt(k)
    bias = rnd((n,), 186, dev, 0.1)
    z0 = rnd((m, n), 187, dev)
    a_ps, w_ps, z_ps = ps_encode(a, k), ps_encode(w, k, n), ps_encode(z0, n)
    z0q = ps_decode(z_ps, n)
    hi_p, l8_p, sc_p = _planes(m, k, dev)
    wh = torch.zeros(lib().ribca_test_mx_weight_bytes(n, k, 0), dtype=torch.uint8, device=dev)
    wx = torch.zeros(lib().ribca_test_mx_weight_bytes(n, k, 1), dtype=torch.uint8, device=dev)
    part = torch.zeros((n // 48, m, 2), dtype=torch.float32, device=dev)
    rs = torch.zeros((m, 2), dtype=torch.float32, device=dev)
    check(lib().ribca_test_gemm_mx_resid(ptr(a_ps), 2 * k, ptr(w_ps), 2 * k, m, n, k, ptr(bias), ptr(hi_p), ptr(l8_p), ptr(sc_p), ptr(wh), ptr(wx),
                                         ptr(z_ps), 2 * n, ptr(part), ptr(rs), None, stream_ptr()), "gemm_mx_resid")
    got = ps_decode(z_ps, n)
    a_hi, a_lo = _ps_halves(a_ps, k)
    w_hi, w_lo = _ps_halves(w_ps, k)
    _, a_lo_q, _, _ = mx.pack_act(a_hi, a_lo.astype(np.float64))
    emu = torch.from_numpy(mx.gemm(a_hi, a_lo_q, w_hi[:n], w_lo[:n])).to(dev)
    scale = 1.0 + (a.double().abs() @ w.double().abs().t())
    e_emu = ((got - (z0q + emu + bias.double())).abs() / scale).max().item()
    e_ref = ((got - (z0q + a.double() @ w.double().t() + bias.double())).abs() / scale).max().item()
    note_err("gemm_mx outlier operands vs emulated operands", e_emu)
    note_err("gemm_mx outlier operands vs exact product", e_ref)
    assert e_emu < 2e-6, e_emu
    # the scheme's worst case (test_gemm_mx_resid): 2^-13 of sum |a w|.  Gaussian operands land at 2-4e-6 (random signs); with Student-t(3)
    # weights the fp6 image of W hi flushes the small weights of a block that holds a 10-30 sigma one, i.e. drops THEIR A lo x W hi term:
    # measured 3.9e-5
    assert e_ref < 2.0 ** -13, e_ref


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
    got = _lo_codes_close(l8_p, l8_r, sc_p, m, f"gelu_mx {m}x{n}")
    # and the operand as fc2 will see it against the exact product
    ln = torch.nn.functional.layer_norm(zq, (d,), g.double(), b.double(), 1e-6)
    ref = torch.nn.functional.gelu(ln @ w.double().t() + bias.double()).cpu().numpy()
    val = _mx3_value(hi_p, l8_p, sc_p, m, n)
    err = np.abs(val - ref).max()
    note_err(f"gelu_mx hi + lo vs exact {m}x{n}x{d} mean {mean}", err)
    # test_gemm_fold_gelu's bound plus the e4m3 rounding of lo: 2^-4 of |lo| <= 2^-16 of the block's largest value
    assert err < 2e-5 * (1.0 + abs(mean) / std) + 2.0 ** -15 * np.abs(ref).max(), err


@pytest.mark.parametrize("m,d", [(1, 384), (150, 384), (260, 576), (300, 192), (5000, 576), (12000, 384)])
@pytest.mark.parametrize("mean,std", [(0.5, 3.0), (30.0, 1.0)])
def test_gemm_mx_fc1(dev, m, d, mean, std):
    """norm2 -> mlp.fc1 folded on the MX kernel (the residual rows read in MX3, K padded to 128: D = 576 -> 640), GELU output in MX3 from
    48-column wave blocks (a 32-column scale block is then shared by two waves): values against the exact product, the scale rule on the
    emitted hi plane, and against the fp16x3 kernel's output"""
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    n = 4 * d
    z_ps, zq, g, b, dp = _ln_case(m, d, 50, dev, mean, std, row_scale=(mean == 0.5))
    w = rnd((n, d), 53, dev, 2.0 / np.sqrt(d))
    bias = rnd((n,), 54, dev, 0.1)
    w_ps, csum, bias2 = _fold(w, g, b, bias, dp, dev)
    rs = _row_stats(z_ps, dp, m, d, dev)
    kz = (dp + 127) // 128 * 128
    a_hi, a_l8, a_sc = _planes(m, kz, dev)
    wh = torch.zeros(lib().ribca_test_mx_weight_bytes(n, kz, 0), dtype=torch.uint8, device=dev)
    wx = torch.zeros(lib().ribca_test_mx_weight_bytes(n, kz, 1), dtype=torch.uint8, device=dev)
    hi_p, l8_p, sc_p = _planes(m, n, dev)
    check(lib().ribca_test_gemm_mx_fc1(ptr(z_ps), 2 * dp, ptr(w_ps), 2 * dp, m, n, dp, ptr(bias2), ptr(csum), ptr(rs), ptr(a_hi), ptr(a_l8), ptr(a_sc),
                                       ptr(wh), ptr(wx), ptr(hi_p), ptr(l8_p), ptr(sc_p), stream_ptr()), "gemm_mx_fc1")
    assert _scale_rule_holds(hi_p, sc_p, m, n)
    val = _mx3_value(hi_p, l8_p, sc_p, m, n)
    lnz = torch.nn.functional.layer_norm(zq, (d,), g.double(), b.double(), 1e-6)
    ref = torch.nn.functional.gelu(lnz @ w.double().t() + bias.double()).cpu().numpy()
    # the product's own error (test_gemm_mx_resid: < 3e-5 of sum_k |a w|, here of the UN-normalised rows times rstd) passes through a GELU
    # of slope <= 1.13; on top of it the bound of test_gemm_gelu_mx
    rstd = rs[:, 0:1].double()
    sum_aw = ((zq.abs() @ (w.double() * g.double()).abs().t()) * rstd).cpu().numpy()
    tol = 2e-5 * (1.0 + abs(mean) / std) + 2.0 ** -15 * np.abs(ref).max() + 3.4e-5 * sum_aw
    err = np.abs(val - ref)
    note_err(f"gemm_mx_fc1 hi + lo vs exact {m}x{n}x{d} mean {mean}", (err / (1.0 + sum_aw)).max())
    assert np.all(err <= tol), (err - tol).max()
    # lo is the e4m3 rounding of x - hi for the kernel's own x: |lo| <= half an ulp of hi (+ one e4m3 step)
    hi = hi_p.cpu().numpy().view(np.float16)[:, mx.hi_pos(np.arange(n))].astype(np.float64)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(hi), 2.0 ** -14))) - 10)
    assert np.all(np.abs(val - hi) <= 0.5 * ulp * (1.0 + 2.0 ** -3) + 2.0 ** -24)
    # against the fp16x3 kernel on the same operands
    out = torch.zeros((m, 2 * n), dtype=torch.int16, device=dev)
    check(lib().ribca_test_gemm_fold(1, ptr(z_ps), 2 * dp, ptr(w_ps), 2 * dp, m, n, dp, ptr(bias2), ptr(csum), ptr(rs), ptr(out), 2 * n,
                                     stream_ptr()), "gemm_fold")
    x3 = ps_decode(out, n).cpu().numpy()
    assert np.all(np.abs(val - x3) <= tol)


@pytest.mark.parametrize("d,cells", [(576, 1), (576, 37), (384, 20), (384, 1)])
@pytest.mark.parametrize("mean,std", [(0.5, 1.0), (30.0, 1.0)])
def test_qkv_attention_mx(dev, d, cells, mean, std):
    """norm1 -> attn.qkv folded on the MX kernel + attention: against LayerNorm + qkv + softmax attention in fp64"""
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    heads, ntok = 12, 101
    hd = d // heads
    hdp = (hd + 7) // 8 * 8
    m = cells * ntok
    z_ps, zq, g, b, dp = _ln_case(m, d, 60, dev, mean, std)
    w = rnd((3 * d, d), 63, dev, 1.0 / np.sqrt(d))
    bias = rnd((3 * d,), 64, dev, 0.1)
    w_ps, csum, bias2 = _fold(w, g, b, bias, dp, dev)
    rs = _row_stats(z_ps, dp, m, d, dev)
    kz = (dp + 127) // 128 * 128
    a_hi, a_l8, a_sc = _planes(m, kz, dev)
    wh = torch.zeros(lib().ribca_test_mx_weight_bytes(3 * d, kz, 0), dtype=torch.uint8, device=dev)
    wx = torch.zeros(lib().ribca_test_mx_weight_bytes(3 * d, kz, 1), dtype=torch.uint8, device=dev)
    q = torch.zeros((cells, heads, 112, 2 * hdp), dtype=torch.int16, device=dev)
    k = torch.zeros_like(q)
    vt = torch.zeros_like(q)
    out = torch.zeros((m, 2 * dp), dtype=torch.int16, device=dev)
    check(lib().ribca_test_qkv_attention_mx(ptr(z_ps), 2 * dp, ptr(w_ps), 2 * dp, cells, d, dp, ptr(bias2), ptr(csum), ptr(rs), ptr(a_hi), ptr(a_l8),
                                            ptr(a_sc), ptr(wh), ptr(wx), ptr(q), ptr(k), ptr(vt), ptr(out), 2 * dp, stream_ptr()), "qkv mx")
    lnz = torch.nn.functional.layer_norm(zq, (d,), g.double(), b.double(), 1e-6)
    qkv = (lnz @ w.double().t() + bias.double()).reshape(cells, ntok, 3, heads, hd).permute(2, 0, 3, 1, 4)
    qq, kk, vv = qkv[0], qkv[1], qkv[2]
    # error scale of a product element: sum_k |z_k (gamma w)_k| * rstd (3e-5 of it, test_gemm_mx_resid) beside the fp16x3 path's bound
    sum_aw = ((zq.abs() @ (w.double() * g.double()).abs().t()) * rs[:, 0:1].double()).reshape(cells, ntok, 3, heads, hd).permute(2, 0, 3, 1, 4)
    amp = 1.0 + abs(mean) / std
    qd = ps_decode(q.reshape(-1, 2 * hdp), hdp).reshape(cells, heads, 112, hdp)
    kd = ps_decode(k.reshape(-1, 2 * hdp), hdp).reshape(cells, heads, 112, hdp)
    assert torch.all((qd[:, :, :ntok, :hd] - qq * hd ** -0.5).abs() <= (qq.abs() + 1.0) * 1e-5 * amp + 3.4e-5 * sum_aw[0] * hd ** -0.5)
    assert torch.all((kd[:, :, :ntok, :hd] - kk).abs() <= (kk.abs() + 1.0) * 1e-5 * amp + 3.4e-5 * sum_aw[1])
    att = torch.softmax(qq @ kk.transpose(-1, -2) * hd ** -0.5, dim=-1) @ vv
    ref = att.permute(0, 2, 1, 3).reshape(m, d)
    err = (ps_decode(out, d) - ref).abs().max().item()
    note_err(f"qkv_attention_mx d={d} cells={cells} mean {mean}", err)
    # the fp16x3 bound of test_qkv_attention_fold + the 2^-16-class error of the MX products on q, k (through the softmax) and v
    assert err < (4e-5 + 2e-4) * amp, err


@pytest.mark.parametrize("kind,m,n,k", [(0, 1, 576, 576), (0, 130, 576, 576), (0, 5000, 576, 576), (0, 700, 384, 384), (0, 300, 192, 192),
                                        (1, 130, 576, 2304), (1, 4000, 576, 2304), (1, 700, 384, 1536), (1, 257, 192, 128)])
@pytest.mark.parametrize("recentre", [False, True])
def test_gemm_resid_zmx(dev, kind, m, n, k, recentre):
    """attn.proj (packed-split operand, two-workgroups kernel) and mlp.fc2 (MX kernel) writing the new residual rows twice: the MX3 copy is
    what the packer makes of the packed-split copy -- hi plane and scale bytes bit for bit, lo bytes up to the double rounding of the
    detour -- and the packed-split copy and statistics are those of the launch without the second copy"""
    from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
    a = rnd((m, k), 5, dev) * torch.exp(rnd((m, 1), 55, dev))
    w = rnd((n, k), 6, dev, 1.0 / np.sqrt(k))
    bias = rnd((n,), 7, dev, 0.1)
    npd = (n + 31) // 32 * 32
    zk = (n + 127) // 128 * 128
    z0 = (rnd((m, n), 8, dev) + (3.0 if recentre else 0.0)) * torch.exp(rnd((m, 1), 58, dev) * 2.0)
    a_ps = ps_encode(a, k)
    w_ps = ps_encode(w, k, lib().ribca_gemm_padded_n(n))
    z_ps = ps_encode(z0, npd)
    z_ref = z_ps.clone()
    prev = _row_stats(z_ps, npd, m, n, dev) if recentre else None
    tiles = n // 48
    part = torch.zeros((tiles, m, 2), dtype=torch.float32, device=dev)
    rs = torch.zeros((m, 2), dtype=torch.float32, device=dev)
    z_hi, z_l8, z_sc = _planes(m, zk, dev)
    a_hi, a_l8, a_sc = _planes(m, k if k % 128 == 0 else 128, dev)
    if kind == 0:
        wsc = torch.zeros_like(w_ps)
        wx = torch.zeros(16, dtype=torch.uint8, device=dev)
    else:
        wsc = torch.zeros(lib().ribca_test_mx_weight_bytes(n, k, 0), dtype=torch.uint8, device=dev)
        wx = torch.zeros(lib().ribca_test_mx_weight_bytes(n, k, 1), dtype=torch.uint8, device=dev)
    check(lib().ribca_test_gemm_resid_zmx(kind, ptr(a_ps), 2 * k, ptr(w_ps), 2 * k, m, n, k, ptr(bias), ptr(a_hi), ptr(a_l8), ptr(a_sc), ptr(wsc), ptr(wx),
                                          ptr(z_ps), 2 * npd, ptr(part), ptr(rs), ptr(prev) if recentre else None, ptr(z_hi), ptr(z_l8), ptr(z_sc), zk,
                                          stream_ptr()), "gemm_resid_zmx")
    # the same launch without the second copy
    part2, rs2 = torch.zeros_like(part), torch.zeros_like(rs)
    if kind == 0:
        wf = torch.zeros_like(w_ps)
        check(lib().ribca_test_gemm_resid_ps_duo(ptr(a_ps), 2 * k, ptr(w_ps), 2 * k, m, n, k, ptr(bias), ptr(wf), ptr(z_ref), 2 * npd, ptr(part2), ptr(rs2),
                                                 ptr(prev) if recentre else None, stream_ptr()), "resid_ps_duo")
    else:
        b_hi, b_l8, b_sc = _planes(m, k, dev)
        check(lib().ribca_test_gemm_mx_resid(ptr(a_ps), 2 * k, ptr(w_ps), 2 * k, m, n, k, ptr(bias), ptr(b_hi), ptr(b_l8), ptr(b_sc), ptr(wsc), ptr(wx),
                                             ptr(z_ref), 2 * npd, ptr(part2), ptr(rs2), ptr(prev) if recentre else None, stream_ptr()), "gemm_mx_resid")
    assert torch.equal(z_ps, z_ref)
    assert torch.equal(rs, rs2)
    # the packer on the packed-split copy
    r_hi, r_l8, r_sc = _planes(m, zk, dev)
    check(lib().ribca_test_mx_pack_act(ptr(z_ps), 2 * npd, m, npd, ptr(r_hi), ptr(r_l8), ptr(r_sc), stream_ptr()), "mx_pack_act")
    assert torch.equal(z_hi, r_hi)
    nb = n // 32
    assert np.array_equal(_scales(z_sc, m)[:, :nb], _scales(r_sc, m)[:, :nb])
    # (the columns of the K pad are never written by the fused producer: the buffers' zeros; the packer writes zeros there -- the permuted
    # hi plane was compared whole above)
    assert torch.all(z_l8[:, n:] == 0)
    _lo_codes_close(z_l8, r_l8, r_sc, m, f"resid_zmx kind {kind} {m}x{n}x{k}")
