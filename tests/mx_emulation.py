"""numpy statement of the MX3 operand format and of the roundings of csrc/gemm_mx.hip (test infrastructure: the GPU tests compare the
packers bit for bit with this and the GEMM with the product of exactly these quantised operands).

Formats (OCP): e4m3 = 4 exponent bits (bias 7), 3 mantissa bits, subnormals, max 448;  e2m3 = 2 exponent bits (bias 1), 3 mantissa bits,
subnormals (step 0.125), max 7.5.  Both conversions round to nearest even; e2m3 saturates, e4m3 is never driven past 256 by the format's
scale rule (tools/mx_mix_probe.hip pins the hardware's behaviour on both).
"""
import numpy as np


def e4m3_round(x):
    """nearest-even e4m3 value of x (float64 array), |x| <= 448"""
    a = np.abs(x)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    e = np.clip(e, -6, 8)
    step = 2.0 ** (e - 3)
    return np.sign(x) * np.rint(a / step) * step


def e2m3_round(x):
    a = np.abs(x)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    e = np.clip(e, 0, 2)
    step = 2.0 ** (e - 3)
    return np.sign(x) * np.minimum(np.rint(a / step) * step, 7.5)


def e4m3_decode(codes):
    c = codes.astype(np.int64)
    s, e, m = (c >> 7) & 1, (c >> 3) & 15, c & 7
    v = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7))
    return np.where(s == 1, -v, v)


def f16_exp_field(x16):
    """exponent field of fp16 values (array of float16)"""
    return ((x16.view(np.uint16) >> 10) & 31).astype(np.int64)


def hi_pos(c):
    """position of logical column c in the permuted hi plane"""
    c = np.asarray(c)
    return (c & ~127) | (((c >> 3) & 3) << 5) | (((c >> 5) & 3) << 3) | (c & 7)


def pack_act(hi16, lo):
    """hi16: float16 [M, Kp128] (already zero-padded), lo: float64 [M, Kp128] -> (hi plane float16 permuted, lo dequantised float64,
    lo as e4m3 VALUES before the scale (for code comparison), scale bytes uint8 [M, Kp128 / 32])"""
    m, k = hi16.shape
    blocks = hi16.reshape(m, k // 32, 32)
    ef = f16_exp_field(np.abs(blocks)).max(axis=2)
    sl = np.maximum(ef, 1) + 93
    scale = 2.0 ** (sl.astype(np.float64) - 127)
    q = e4m3_round(lo.reshape(m, k // 32, 32) / scale[:, :, None])
    plane = np.zeros_like(hi16)
    plane[:, hi_pos(np.arange(k))] = hi16
    return plane, (q * scale[:, :, None]).reshape(m, k), q.reshape(m, k), sl.astype(np.uint8)


def fp6_image(x16):
    """x16 float16 [R, K]: the fp6 image of every 32-column block scaled so that its largest magnitude lands in [4, 8) (weights: both hi
    and lo get their own exponent; activations' hi: the same rule) -> dequantised float64"""
    r, k = x16.shape
    blocks = x16.reshape(r, k // 32, 32)
    ef = np.maximum(f16_exp_field(np.abs(blocks)).max(axis=2), 1)
    scale = 2.0 ** (ef.astype(np.float64) - 15 - 2)
    q = e2m3_round(blocks.astype(np.float64) / scale[:, :, None])
    return (q * scale[:, :, None]).reshape(r, k)


def gemm(a_hi16, a_lo_q, w_hi16, w_lo16):
    """the three products of gemm_mx.hip in float64: Ah Wh^T + Al' Wh'^T + Ah' Wl'^T  (a_lo_q: the dequantised e4m3 lo of pack_act)"""
    ah, wh = a_hi16.astype(np.float64), w_hi16.astype(np.float64)
    return ah @ wh.T + a_lo_q @ fp6_image(w_hi16).T + fp6_image(a_hi16) @ fp6_image(w_lo16).T
