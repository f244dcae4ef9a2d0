"""Precision study for the matrix-core operand formats of the ViT forward (test infrastructure; not collected by pytest).

Emulates, on the CPU oracle ViT (``oracle.ref_vit``), what each candidate operand format of the HIP GEMM / attention
kernels does to the per-cell softmax confidences, against the plain fp32 forward (the reference's CPU path,
``cell_type_annotation/model.py:397-406``).  Every matrix product  C = A . B^T  of the block (qkv, Q.K^T, P.V, proj, fc1,
fc2) is replaced by the scheme's sum of reduced-precision products, accumulated in fp32; the residual stream, LayerNorm,
softmax and GELU stay fp32, as in the kernels.

    python tests/precision_study.py --cells 256 --models immune_base immune_full --schemes bf16x3 f16_bf8t ...

Schemes (x = h + l:  h = round16(x), l = x - h;  h' = low-precision copy of h used only against the other side's l):
    bf16        h.h                      h = bf16                       1 MFMA pass
    f16         h.h                      h = fp16                       1 pass
    bf16x3      h.h + l.h + h.l          h, l = bf16                    3 passes   (round-1 production path)
    f16x3       h.h + l.h + h.l          h, l = fp16                    3 passes
    f16_bf8t    h.h + l8.h' + h'.l8      h = fp16, h' = high byte of h (e5m2, truncated), l8 = e5m2(2^12 l)    1 + 2 x 1/2
    f16_bf8r    same, h' rounded to nearest e5m2
    f16_fp8     same, h' = e4m3(h) l8 = e4m3(2^12 l)
    f16_fp6     same, h' and l as MX e2m3 with one power-of-two scale per 32 k                                 1 + 2 x 1/4
    f16_fp6t    fp6 l, h' = high byte (e5m2 truncation)  -- mixed fp8 x fp6 MFMA (fp8 rate)
Round 5 (spend the confidence headroom) -- the MX mix of csrc/gemm_mx.hip and its one-correction reductions; a = activation,
w = weight; the attention products and every linear not named by --linears stay f16x3:
    mx175       ah.wh + al8.wh6 + ah6.wl6   al8 = e4m3 of l under the MX3 block scale, ?6 = MX e2m3             1 + 1/2 + 1/4   (production)
    mx150       ah.wh + al8.wh6             no W lo image at all (weights rounded to fp16)                       1 + 1/2
    mx125       ah.wh + ah6.wl6             no A lo plane at all (activations rounded to fp16: 2 bytes / element) 1 + 1/4
    mx100       ah.wh                       one fp16 pass                                                        1
    mx150s      ah.wh + al6.wh6 + ah6.wl6   BOTH corrections kept, A lo as MX e2m3 under its own block scale (the fp8 operand is what makes
                                            the first correction a half-rate instruction): 2.78 bytes / element       1 + 1/4 + 1/4
--zstream: the format the residual stream z is STORED in between the residual GEMMs (fp32: not rounded, the study's default; ps: fp16 hi + fp16
lo, what the kernels keep; mx3: fp16 hi + e4m3 lo under the MX3 block scale -- 3 bytes per element, the stream attn.proj / mlp.fc2 would read
and write if the packed-split copy were dropped).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiplexed_image_annotator_amd import synth  # noqa: E402
from oracle import ref_vit  # noqa: E402

LO_SCALE = 4096.0  # 2^12: l of an fp16-rounded value is below 2^-11 |x|
ZSTREAM = "fp32"


def store_z(z):
    """the residual stream as it would sit in memory between two residual GEMMs (--zstream)"""
    if ZSTREAM == "fp32":
        return z
    h = r_f16(z)
    if ZSTREAM == "ps":
        return h + r_f16(z - h)
    return h + mx3_lo(z, h)


def r_bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def r_f16(x):
    return x.to(torch.float16).to(torch.float32)


def r_e5m2(x):
    return x.clamp(-57344.0, 57344.0).to(torch.float8_e5m2).to(torch.float32)


def r_e4m3(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def t_e5m2_of_f16(h):
    """high byte of the fp16 encoding = the value truncated (toward zero) to e5m2"""
    bits = h.to(torch.float16).view(torch.int16)
    return (bits & -256).view(torch.float16).to(torch.float32)


def mx_e2m3(x, block=32):
    """OCP MX fp6 (e2m3): per ``block`` consecutive k one power-of-two scale 2^(floor(log2 max) - 2); elements rounded to
    nearest on the e2m3 grid (subnormal step 1/8, max 7.5, saturating)."""
    k = x.shape[-1]
    pad = (-k) % block
    xp = F.pad(x, (0, pad)) if pad else x
    xb = xp.reshape(*xp.shape[:-1], -1, block)
    amax = xb.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(torch.where(amax > 0, amax, torch.ones_like(amax)))) - 2.0
    scale = torch.exp2(e)
    y = (xb / scale).clamp(-7.5, 7.5)
    ay = y.abs()
    ex = torch.floor(torch.log2(torch.where(ay >= 1.0, ay, torch.ones_like(ay))))      # 0, 1, 2 for normals; subnormals use 0
    step = torch.exp2(ex - 3.0)
    q = torch.round(ay / step) * step        # round half to even on the grid
    q = torch.sign(y) * q.clamp(max=7.5)
    out = (q * scale).reshape(*xp.shape)
    return out[..., :k] if pad else out


def split(x, scheme):
    """-> (h, hp, l): main operand, its low-precision copy, low-precision remainder (all fp32 values)."""
    if scheme == "bf16":
        return r_bf16(x), None, None
    if scheme == "f16":
        return r_f16(x), None, None
    if scheme == "bf16x3":
        h = r_bf16(x)
        return h, h, r_bf16(x - h)
    if scheme == "f16x3":
        h = r_f16(x)
        return h, h, r_f16(x - h)
    h = r_f16(x)
    l = x - h
    if scheme == "f16_bf8t":
        return h, t_e5m2_of_f16(h), r_e5m2(l * LO_SCALE) / LO_SCALE
    if scheme == "f16_bf8r":
        return h, r_e5m2(h), r_e5m2(l * LO_SCALE) / LO_SCALE
    if scheme == "f16_fp8":
        return h, r_e4m3(h), r_e4m3(l * LO_SCALE) / LO_SCALE
    if scheme == "f16_bf8t_fp8":        # h' = truncated high byte, l = e4m3
        return h, t_e5m2_of_f16(h), r_e4m3(l * LO_SCALE) / LO_SCALE
    if scheme == "f16_fp6":
        return h, mx_e2m3(h), mx_e2m3(l)
    if scheme == "f16_fp6t":
        return h, t_e5m2_of_f16(h), mx_e2m3(l)
    raise ValueError(scheme)


# Per-tensor allocation study: scheme "alloc:<spec>" with <spec> = comma list of tensor=format, tensors x1 wqkv q k v p o wproj x2 w1
# hh w2 (default format "full"), formats:  full = f16 h + e5m2 l (3 bytes), used against the other side's exact h;
# h = f16 only (2 bytes);  full4 = f16 h + e4m3 l;  full16 = f16 h + f16 l.
ALLOC = {}


def split_alloc(x, fmt):
    h = r_f16(x)
    if fmt == "h":
        return h, None
    l = x - h
    if fmt == "full":
        return h, r_e5m2(l * LO_SCALE) / LO_SCALE
    if fmt == "full4":
        return h, r_e4m3(l * LO_SCALE) / LO_SCALE
    if fmt == "full16":
        return h, r_f16(l)
    raise ValueError(fmt)


def mm_alloc(a, b, ta, tb):
    ah, al = split_alloc(a, ALLOC.get(ta, "full"))
    bh, bl = split_alloc(b, ALLOC.get(tb, "full"))
    aa, bb = [ah], [bh]
    if al is not None:
        aa.append(al)
        bb.append(bh)
    if bl is not None:
        aa.append(ah)
        bb.append(bl)
    return torch.cat(aa, dim=-1) @ torch.cat(bb, dim=-1).transpose(-2, -1)


def mx3_lo(x, h, block=32):
    """e4m3 image of l = x - h under the MX3 scale rule of csrc/gemm_mx.hip: per 32 k, scale 2^(E - 19) with E the exponent of the block's
    largest |h| (lo / scale <= 256 < 448)"""
    k = x.shape[-1]
    pad = (-k) % block
    l = x - h
    hp, lp = (F.pad(h, (0, pad)), F.pad(l, (0, pad))) if pad else (h, l)
    hb, lb = hp.reshape(*hp.shape[:-1], -1, block), lp.reshape(*lp.shape[:-1], -1, block)
    amax = hb.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(torch.where(amax > 0, amax, torch.full_like(amax, 2.0 ** -14)))).clamp(min=-14.0)
    scale = torch.exp2(e - 19.0)
    out = (r_e4m3(lb / scale) * scale).reshape(*hp.shape)
    return out[..., :k] if pad else out


MX_TERMS = {"mx175": (True, True), "mx150": (True, False), "mx125": (False, True), "mx100": (False, False), "mx150s": (True, True),
            # A lo in a 6- or 4-bit block format whose scale is DERIVED from the block's hi exponent (no second scale byte in the MX3 row):
            "mx150h": (True, True), "mx_a4": (True, True), "mx_a4h": (True, True), "mx_a4w4": (True, True)}
LINEARS = {"wqkv", "wproj", "w1", "w2"}      # --linears: which weights take an mx* scheme (the others: f16x3)


def mx_block(x, block, fmt, ref=None, ref_shift=0.0):
    """block-scaled image of x: fmt e2m3 (max 7.5, 3 mantissa bits) or e2m1 (fp4: 0 .5 1 1.5 2 3 4 6); scale 2^(floor(log2 max|x|) - 2), or --
    ref given -- 2^(floor(log2 max|ref|) + ref_shift): the scale a kernel can derive from the block's hi exponent without a byte of its own"""
    k = x.shape[-1]
    pad = (-k) % block
    xp = F.pad(x, (0, pad)) if pad else x
    xb = xp.reshape(*xp.shape[:-1], -1, block)
    if ref is None:
        amax = xb.abs().amax(dim=-1, keepdim=True)
        e = torch.floor(torch.log2(torch.where(amax > 0, amax, torch.ones_like(amax)))) - 2.0
    else:
        rp = F.pad(ref, (0, pad)) if pad else ref
        amax = rp.reshape(*rp.shape[:-1], -1, block).abs().amax(dim=-1, keepdim=True)
        e = torch.floor(torch.log2(torch.where(amax > 0, amax, torch.full_like(amax, 2.0 ** -14)))).clamp(min=-14.0) + ref_shift
    scale = torch.exp2(e)
    y = xb / scale
    ay = y.abs()
    if fmt == "e2m3":
        ay = ay.clamp(max=7.5)
        ex = torch.floor(torch.log2(torch.where(ay >= 1.0, ay, torch.ones_like(ay))))
        step = torch.exp2(ex - 3.0)
        q = (torch.round(ay / step) * step).clamp(max=7.5)
    else:      # e2m1: subnormal step 0.5 below 1, one mantissa bit above
        ay = ay.clamp(max=6.0)
        ex = torch.floor(torch.log2(torch.where(ay >= 1.0, ay, torch.ones_like(ay))))
        step = torch.exp2(ex - 1.0)
        q = (torch.round(ay / step) * step).clamp(max=6.0)
    out = (torch.sign(y) * q * scale).reshape(*xp.shape)
    return out[..., :k] if pad else out


def mm_mx(a, b, scheme):
    use_al, use_wl = MX_TERMS[scheme]
    ah, bh = r_f16(a), r_f16(b)
    aa, bb = [ah], [bh]
    if use_al:
        # |lo| <= half an ulp of the block's largest hi = 2^(E - 11): scale 2^(E - 13) puts it at <= 4 (e2m3 max 7.5, e2m1 max 6)
        al = {"mx150s": lambda: mx_e2m3(a - ah), "mx150h": lambda: mx_block(a - ah, 32, "e2m3", ah, -13.0),
              "mx_a4": lambda: mx_block(a - ah, 32, "e2m1"), "mx_a4h": lambda: mx_block(a - ah, 32, "e2m1", ah, -13.0),
              "mx_a4w4": lambda: mx_block(a - ah, 32, "e2m1", ah, -13.0)}.get(scheme, lambda: mx3_lo(a, ah))()
        aa.append(al)
        bb.append(mx_e2m3(bh))
    if use_wl:
        aa.append(mx_e2m3(ah))
        bb.append(mx_block(r_f16(b - bh), 32, "e2m1") if scheme == "mx_a4w4" else mx_e2m3(r_f16(b - bh)))
    return torch.cat(aa, dim=-1) @ torch.cat(bb, dim=-1).transpose(-2, -1)


def mm(a, b, scheme, ta=None, tb=None):
    """a (.., m, k) . b (.., n, k)^T under ``scheme``."""
    if scheme in MX_TERMS:
        if tb in LINEARS:
            return mm_mx(a, b, scheme)
        scheme = "f16x3"
    if scheme.startswith("alloc"):
        return mm_alloc(a, b, ta, tb)
    if scheme == "fp32":
        return a @ b.transpose(-2, -1)
    ah, ahp, al = split(a, scheme)
    bh, bhp, bl = split(b, scheme)
    if al is None:
        return ah @ bh.transpose(-2, -1)
    aa = torch.cat((ah, ahp, al), dim=-1)
    bb = torch.cat((bh, bl, bhp), dim=-1)
    return aa @ bb.transpose(-2, -1)


@torch.no_grad()
def logits_scheme(sd, x, scheme, attn_scheme=None):
    """oracle.ref_vit.logits with every block product under ``scheme`` (patch embedding and head stay fp32 as in the kernels:
    embed_f32_kernel / head_softmax_kernel)."""
    attn_scheme = attn_scheme or ("f16x3" if scheme in MX_TERMS else scheme)
    heads = ref_vit.HEADS
    b = x.shape[0]
    d = sd["cls_token"].shape[-1]
    hd = d // heads
    t = ref_vit.patch_embed(sd, x)
    z = store_z(torch.cat((sd["cls_token"].expand(b, -1, -1), t), dim=1) + sd["pos_embed"])
    n = z.shape[1]
    for i in range(ref_vit.depth_of(sd)):
        p = f"blocks.{i}."
        y = F.layer_norm(z, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], ref_vit.LN_EPS)
        qkv = mm(y, sd[p + "attn.qkv.weight"], scheme, "x1", "wqkv") + sd[p + "attn.qkv.bias"]
        qkv = qkv.reshape(b, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
        att = mm(q, k, attn_scheme, "q", "k").softmax(dim=-1)
        y = mm(att, v.transpose(-2, -1), attn_scheme, "p", "v").transpose(1, 2).reshape(b, n, d)
        z = store_z(z + mm(y, sd[p + "attn.proj.weight"], scheme, "o", "wproj") + sd[p + "attn.proj.bias"])
        y = F.layer_norm(z, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], ref_vit.LN_EPS)
        y = F.gelu(mm(y, sd[p + "mlp.fc1.weight"], scheme, "x2", "w1") + sd[p + "mlp.fc1.bias"])
        z = store_z(z + mm(y, sd[p + "mlp.fc2.weight"], scheme, "hh", "w2") + sd[p + "mlp.fc2.bias"])
    z = F.layer_norm(z, (d,), sd["norm.weight"], sd["norm.bias"], ref_vit.LN_EPS)
    return z[:, 0] @ sd["head.weight"].t() + sd["head.bias"]


def patch_like_inputs(name, n, seed):
    """same generator as tests/test_oracle_vit.py: background -1, sparse positive signal"""
    d, c, k = synth.VIT_CONFIGS[name]
    u = synth.uniform(synth.stream_key(seed, "vitx/" + name), n * c * 1600).reshape(n, c, 40, 40).to(torch.float32)
    x = u * 2 - 1
    return torch.where(x > 0.1, x, torch.full_like(x, -1.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--models", nargs="+", default=["immune_base", "immune_full"])
    ap.add_argument("--schemes", nargs="+", default=["bf16x3", "f16", "f16_bf8t", "f16_bf8r", "f16_fp8", "f16_fp6"])
    ap.add_argument("--seed", type=int, default=synth.SEED_BASE + 7)
    ap.add_argument("--linears", nargs="+", default=["wqkv", "wproj", "w1", "w2"], help="weights that take an mx* scheme (others: f16x3)")
    ap.add_argument("--family", choices=sorted(synth.WEIGHT_FAMILIES), default="uniform",
                    help="synthetic weight family: uniform (Xavier-uniform, LN gains near 1) or heavy (Student-t(3) weights, LN gains up to 5, outlier residual channels)")
    ap.add_argument("--zstream", choices=["fp32", "ps", "mx3"], default="fp32", help="storage format of the residual stream between residual GEMMs")
    args = ap.parse_args()
    global ZSTREAM
    ZSTREAM = args.zstream
    LINEARS.clear()
    LINEARS.update(args.linears)
    torch.set_num_threads(os.cpu_count() or 1)
    for name in args.models:
        sd = synth.WEIGHT_FAMILIES[args.family](name, args.seed)
        x = patch_like_inputs(name, args.cells, args.seed + 1)
        t0 = time.time()
        ref = torch.cat([F.softmax(ref_vit.logits(sd, x[i:i + args.batch]), dim=1) for i in range(0, args.cells, args.batch)])
        srt = ref.sort(dim=1, descending=True).values
        margin = (srt[:, 0] - srt[:, 1])
        print(f"{name} [{args.family}]: {args.cells} cells, fp32 reference {time.time() - t0:.1f} s; top-2 margin min {margin.min():.2e} "
              f"median {margin.median():.3f}", flush=True)
        # the reference's own distance from exact arithmetic: what no other correct evaluation can be expected to reproduce
        sd64 = {k_: v_.to(torch.float64) for k_, v_ in sd.items()}
        n64 = min(args.cells, 128)
        ref64 = torch.cat([F.softmax(ref_vit.logits(sd64, x[i:i + args.batch].to(torch.float64)), dim=1) for i in range(0, n64, args.batch)])
        d64 = (ref[:n64].to(torch.float64) - ref64).abs()
        print(f"  fp32 vs fp64   max|dp| {d64.max():.2e}  mean|dp| {d64.mean():.2e}   ({n64} cells)", flush=True)
        for scheme in args.schemes:
            t0 = time.time()
            if scheme.startswith("alloc"):
                ALLOC.clear()
                spec = scheme.split(":", 1)[1] if ":" in scheme else ""
                for item in filter(None, spec.split(",")):
                    k_, v_ = item.split("=")
                    for name_ in k_.split("+"):
                        ALLOC[name_] = v_
            got = torch.cat([F.softmax(logits_scheme(sd, x[i:i + args.batch], scheme), dim=1)
                             for i in range(0, args.cells, args.batch)])
            dp = (got - ref).abs()
            flips = int((got.argmax(1) != ref.argmax(1)).sum())
            print(f"  {scheme:14s} max|dp| {dp.max():.2e}  mean|dp| {dp.mean():.2e}  p99.9 {dp.flatten().quantile(0.999):.2e}  "
                  f"label flips {flips}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
