"""CPU tests of the MX3 operand format and of the MX product's arithmetic as tests/mx_emulation.py states them (the GPU tests compare the
kernels bit for bit with that module; these tests pin the module itself: the formats against exhaustive code tables, the format's
invariants, and the scheme's error against the exact product -- the bound the GPU test asserts, derived here without a GPU)."""
import numpy as np
import pytest

import mx_emulation as mx


def _e4m3_table():
    codes = np.arange(256, dtype=np.uint8)
    vals = mx.e4m3_decode(codes)
    keep = (codes & 0x7f) != 0x7f                     # 0x7f / 0xff are the NaN codes
    return np.unique(vals[keep])


def _e2m3_table():
    out = []
    for s in (1.0, -1.0):
        for e in range(4):
            for m in range(8):
                out.append(s * (m / 8.0 if e == 0 else (1.0 + m / 8.0) * 2.0 ** (e - 1)))
    return np.unique(np.array(out))


def test_e4m3_round_is_nearest_even_on_the_code_table():
    """every representable value is a fixed point; between two neighbours the nearer one wins and the midpoint goes to the even code"""
    t = _e4m3_table()
    assert t.max() == 448.0 and t.min() == -448.0 and 2.0 ** -9 in t
    np.testing.assert_array_equal(mx.e4m3_round(t), t)
    pos = t[t >= 0]
    lo, hi = pos[:-1], pos[1:]
    np.testing.assert_array_equal(mx.e4m3_round(lo + 0.25 * (hi - lo)), lo)
    np.testing.assert_array_equal(mx.e4m3_round(lo + 0.75 * (hi - lo)), hi)
    mid = mx.e4m3_round(0.5 * (lo + hi))
    # index parity in the sorted non-negative table = parity of the code's mantissa LSB
    even = np.where(np.arange(len(lo)) % 2 == 0, lo, hi)
    np.testing.assert_array_equal(mid, even)


def test_e2m3_round_saturates_and_matches_the_code_table():
    t = _e2m3_table()
    assert t.max() == 7.5 and 0.125 in t
    np.testing.assert_array_equal(mx.e2m3_round(t), t)
    assert mx.e2m3_round(np.array([7.8, 100.0, -1e6])).tolist() == [7.5, 7.5, -7.5]
    x = np.linspace(-7.5, 7.5, 4001)
    r = mx.e2m3_round(x)
    assert np.all(np.isin(r, t))
    nearest = t[np.abs(x[:, None] - t[None, :]).argmin(axis=1)]
    ties = np.isclose(np.abs(x[:, None] - t[None, :]).min(axis=1) * 2, np.diff(t).min()) | (np.abs(r - x) == np.abs(nearest - x))
    assert np.all((r == nearest) | ties)


def test_hi_plane_permutation():
    """column c = 32 g + 8 s + j of a 128-column group sits at 32 s + 8 g + j: a permutation of every group, identity across groups"""
    c = np.arange(640)
    p = mx.hi_pos(c)
    assert np.array_equal(np.sort(p), c)
    assert np.array_equal(p // 128, c // 128)
    g, s, j = (c % 128) // 32, (c % 32) // 8, c % 8
    assert np.array_equal(p % 128, 32 * s + 8 * g + j)
    # what the layout is for: lane group g of sub-step s reads 8 contiguous elements, a lane's four sub-steps are 32 consecutive columns
    for grp in range(4):
        cols = np.concatenate([np.where((p % 128 >= 32 * sub + 8 * grp) & (p % 128 < 32 * sub + 8 * grp + 8) & (c < 128))[0] for sub in range(4)])
        assert np.array_equal(cols, np.arange(32 * grp, 32 * grp + 32))


@pytest.mark.parametrize("spread", [1.0, 1e-3, 3e3])
def test_pack_act_invariants(spread):
    """scale rule (lo / scale never reaches the e4m3 range limit: the hardware conversion does not saturate), reconstruction error of
    hi + lo' against x (2^-4 of |lo| <= 2^-16 of the block's largest magnitude), zero and subnormal blocks"""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((64, 256)) * spread * np.exp(rng.standard_normal((64, 1)))).astype(np.float32)
    x[1, :32] = 0.0
    x[2] *= 1e-7 / spread
    x = np.clip(x, -65504, 65504)
    hi = x.astype(np.float16)
    lo = x.astype(np.float64) - hi.astype(np.float64)
    plane, lo_q, q, sl = mx.pack_act(hi, lo)
    assert np.abs(q).max() <= 256.0
    assert sl.min() >= 94 and sl.max() <= 124
    assert np.array_equal(plane[:, mx.hi_pos(np.arange(256))], hi)
    blockmax = np.abs(hi.astype(np.float64)).reshape(64, 8, 32).max(axis=2)
    err = np.abs(hi.astype(np.float64) + lo_q - x.astype(np.float64)).reshape(64, 8, 32).max(axis=2)
    # lo <= half an ulp of hi <= 2^-11 blockmax; e4m3 keeps 4 significant bits (relative 2^-4); subnormal e4m3 step 2^-9 of the scale
    scale = 2.0 ** (sl.astype(np.float64) - 127)
    assert np.all(err <= 2.0 ** -15 * np.maximum(blockmax, 2.0 ** -14) + scale * 2.0 ** -10)


@pytest.mark.parametrize("m,n,k", [(24, 48, 128), (16, 96, 640), (8, 48, 2304)])
def test_scheme_error_against_exact_product(m, n, k):
    """fp16 hi * hi + (A lo as e4m3) x (W hi as e2m3) + (A hi as e2m3) x (W lo as e2m3), every primed operand block-scaled per 32 k:
    each correction operand keeps 4 significant bits (relative error <= 2^-4) and multiplies a factor <= 2^-11 of the product, and the
    lo * lo term (2^-22) is dropped: |error| <= sum_k |a w| * (2 * 2^-15 + 2^-22) in the worst case -- the GPU test's 3e-5 bound --
    and a random walk in practice"""
    rng = np.random.default_rng(11)
    a = (rng.standard_normal((m, k)) * np.exp(rng.standard_normal((m, 1)))).astype(np.float32)
    a = np.where(rng.random((m, k)) > 0.5, a * 8.0, a).astype(np.float32)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    a_hi, w_hi = a.astype(np.float16), w.astype(np.float16)
    a_lo = a.astype(np.float64) - a_hi.astype(np.float64)
    w_lo16 = (w.astype(np.float64) - w_hi.astype(np.float64)).astype(np.float16)
    _, a_lo_q, _, _ = mx.pack_act(a_hi, a_lo)
    got = mx.gemm(a_hi, a_lo_q, w_hi, w_lo16)
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    scale = np.abs(a.astype(np.float64)) @ np.abs(w.astype(np.float64)).T
    rel = (np.abs(got - exact) / scale).max()
    assert rel < 3e-5, rel
    assert rel > 1e-8          # the scheme is not exact: a zero here would mean the emulation dropped a rounding
    # hi * hi alone is 2^-11 class per term (a random walk over k): the corrections buy more than a decimal digit
    rel_hh = (np.abs(a_hi.astype(np.float64) @ w_hi.astype(np.float64).T - exact) / scale).max()
    assert rel_hh > 8 * rel
