"""Deterministic synthetic tiles, masks and weights for tests and ``bench.py``.

The reference ships no weights (``download_models.py:7-24`` pulls them from the network) and
its example images are missing blobs, so every parity and throughput input is synthetic
(SURVEY.md §8d).  Everything here is produced by a counter-based integer hash evaluated with
torch int64 arithmetic, which is bit-exact on CPU and GPU and independent of any library RNG:
the build container, the GPU box and every rank generate identical bytes.

No transcendental functions are used (their last ulp may differ between machines): shapes are
rational bumps, "normal" draws are sums of four uniforms.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

SEED_BASE = 0x524942430000  # "RIBC" << 16, config k uses SEED_BASE + k

_M64 = (1 << 64) - 1


def _s64(v: int) -> int:
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


_C1 = _s64(0xBF58476D1CE4E5B9)
_C2 = _s64(0x94D049BB133111EB)
_GOLD = _s64(0x9E3779B97F4A7C15)


def _srl(x: torch.Tensor, s: int) -> torch.Tensor:
    return (x >> s) & ((1 << (64 - s)) - 1)


def mix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (two's complement wrap-around)."""
    x = (x ^ _srl(x, 30)) * _C1
    x = (x ^ _srl(x, 27)) * _C2
    return x ^ _srl(x, 31)


def fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & _M64
    return h


def _mix_py(v: int) -> int:
    v &= _M64
    v = ((v ^ (v >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    v = ((v ^ (v >> 27)) * 0x94D049BB133111EB) & _M64
    return v ^ (v >> 31)


def stream_key(seed: int, name: str, k: int = 0) -> int:
    return _s64(_mix_py(seed ^ _mix_py(fnv1a(name) + k * 0x9E3779B97F4A7C15)))


def hash_u24(key: int, idx: torch.Tensor) -> torch.Tensor:
    """24 uniform bits per counter (int64 tensor in [0, 2^24))."""
    return _srl(mix64(idx * _GOLD + key), 40)


def uniform(key: int, n: int, device="cpu") -> torch.Tensor:
    """float64 uniforms in [0,1) on a 2^-24 grid (exactly representable in fp32)."""
    idx = torch.arange(n, dtype=torch.int64, device=device)
    return hash_u24(key, idx).to(torch.float64) / float(1 << 24)


def approx_normal(key: int, n: int, device="cpu") -> torch.Tensor:
    """Sum of four uniforms, centred and scaled to unit variance (exact in fp64)."""
    idx = torch.arange(n, dtype=torch.int64, device=device) * 4
    s = torch.zeros(n, dtype=torch.int64, device=device)
    for j in range(4):
        s += hash_u24(key, idx + j)
    return (s.to(torch.float64) / float(1 << 24) - 2.0) * math.sqrt(3.0)


# ----------------------------------------------------------------------------------------------
# ViT classifier weights (timm state-dict keys)
# ----------------------------------------------------------------------------------------------

#: name -> (embed dim D, input channels C, classes K); reference CTA/model.py:66-88,188-234
VIT_CONFIGS: Dict[str, Tuple[int, int, int]] = {
    "nerve": (144, 3, 2),
    "immune_base": (288, 7, 5),
    "struct": (288, 7, 6),
    "immune_extended": (384, 10, 8),
    "immune_full": (576, 15, 12),
}
VIT_DEPTH = 12
VIT_HEADS = 12
VIT_TOKENS = 101  # 10x10 patches of 4x4 px + CLS


def vit_flops_per_cell(name: str) -> float:
    """Algorithmic FLOPs per cell, BASELINE.md §3 formula."""
    d, c, k = VIT_CONFIGS[name]
    n = VIT_TOKENS
    return 2.0 * 100 * 16 * c * d + VIT_DEPTH * (24.0 * n * d * d + 4.0 * n * n * d) + 2.0 * d * k


def make_vit_state_dict(name: str, seed: int, depth: int = VIT_DEPTH, head_gain: float = 4.0, lin_gain: float = 2.0,
                        pe_gain: float = 4.0) -> Dict[str, torch.Tensor]:
    """Seeded random fp32 state dict with the keys a timm ``VisionTransformer`` checkpoint has.

    Linear weights are Xavier-uniform x ``lin_gain``, embeddings ~N(0, 0.02^2), LayerNorm gains/biases
    and linear biases are perturbed away from (1, 0) so that a kernel that drops one is caught.
    The patch-embed bias cancels the all -1 background (``bias += sum(W)``) so that background tokens
    embed to ~pos_embed only and the per-cell signal survives the first LayerNorm: outputs then depend
    on the cell instead of collapsing to one class, while the net stays in the non-chaotic regime of a
    trained model (gains above ~3 make a random ViT amplify rounding noise unrealistically).
    ``head.weight`` ~ N(0, (head_gain/sqrt(D))^2) gives logits with std ~3-4 (decisive softmax).
    """
    d, c, k = VIT_CONFIGS[name]
    sd: Dict[str, torch.Tensor] = {}

    def nrm(key_name, shape, std):
        n = int(math.prod(shape))
        return (approx_normal(stream_key(seed, name + "/" + key_name), n) * std).to(torch.float32).reshape(shape)

    def uni(key_name, shape, lo, hi):
        n = int(math.prod(shape))
        return (uniform(stream_key(seed, name + "/" + key_name), n) * (hi - lo) + lo).to(torch.float32).reshape(shape)

    def xavier(key_name, out_f, in_f):
        a = lin_gain * math.sqrt(6.0 / (in_f + out_f))
        return uni(key_name, (out_f, in_f), -a, a)

    sd["cls_token"] = nrm("cls_token", (1, 1, d), 0.02)
    sd["pos_embed"] = nrm("pos_embed", (1, VIT_TOKENS, d), 0.02)
    a = pe_gain * math.sqrt(6.0 / (16 * c + d))
    sd["patch_embed.proj.weight"] = uni("patch_embed.proj.weight", (d, c, 4, 4), -a, a)
    sd["patch_embed.proj.bias"] = (uni("patch_embed.proj.bias", (d,), -0.02, 0.02)
                                   + sd["patch_embed.proj.weight"].to(torch.float64).sum(dim=(1, 2, 3)).to(torch.float32))
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = uni(p + "norm1.weight", (d,), 0.9, 1.1)
        sd[p + "norm1.bias"] = uni(p + "norm1.bias", (d,), -0.05, 0.05)
        sd[p + "attn.qkv.weight"] = xavier(p + "attn.qkv.weight", 3 * d, d)
        sd[p + "attn.qkv.bias"] = uni(p + "attn.qkv.bias", (3 * d,), -0.02, 0.02)
        sd[p + "attn.proj.weight"] = xavier(p + "attn.proj.weight", d, d)
        sd[p + "attn.proj.bias"] = uni(p + "attn.proj.bias", (d,), -0.02, 0.02)
        sd[p + "norm2.weight"] = uni(p + "norm2.weight", (d,), 0.9, 1.1)
        sd[p + "norm2.bias"] = uni(p + "norm2.bias", (d,), -0.05, 0.05)
        sd[p + "mlp.fc1.weight"] = xavier(p + "mlp.fc1.weight", 4 * d, d)
        sd[p + "mlp.fc1.bias"] = uni(p + "mlp.fc1.bias", (4 * d,), -0.02, 0.02)
        sd[p + "mlp.fc2.weight"] = xavier(p + "mlp.fc2.weight", d, 4 * d)
        sd[p + "mlp.fc2.bias"] = uni(p + "mlp.fc2.bias", (d,), -0.02, 0.02)
    sd["norm.weight"] = uni("norm.weight", (d,), 0.9, 1.1)
    sd["norm.bias"] = uni("norm.bias", (d,), -0.05, 0.05)
    sd["head.weight"] = nrm("head.weight", (k, d), head_gain / math.sqrt(d))
    sd["head.bias"] = uni("head.bias", (k,), -0.1, 0.1)
    return sd


def make_vit_state_dict_heavy(name: str, seed: int, depth: int = VIT_DEPTH, head_gain: float = 4.0, lin_gain: float = 1.5,
                              gamma_max: float = 5.0, outlier_channels: int = 4, outlier_gain: float = 50.0) -> Dict[str, torch.Tensor]:
    """A second weight family for the parity audits (same keys as ``make_vit_state_dict``): what trained ViTs have and the uniform
    family lacks.  No checkpoint of the reference can be loaded here (download_models.py:7-24 needs the network), so this is the stand-in
    for model.py:188-239's real weights:

    * linear weights Student-t(3), unit variance x Xavier std x ``lin_gain``: heavy tails -- single weights 10-30 sigma out, so a 32-wide
      block of the MX weight image holds one value far above the rest (the small ones lose bits under the shared block scale);
    * LayerNorm gains log-uniform in [1 / gamma_max, gamma_max]: per-channel scales two decades apart inside the folded weight gamma o W;
    * ``outlier_channels`` residual channels carry ``outlier_gain`` x the others through ``pos_embed`` / ``cls_token``: massive
      activations -- one |x| >> the rest inside a 32-column MX3 block of the residual rows, rows with |mean| / std >> 1.

    ``lin_gain`` = 1.5 (the uniform family: 2) offsets the larger LayerNorm gains (rms 2.3 instead of 1) so that the net stays in the
    non-chaotic regime of a trained model: on real patches (mostly background) the fp32 CPU forward itself then sits 1-8e-5 from the fp64
    forward, as with the uniform family (8e-5); at lin_gain 2 the reference's OWN fp32 rounding noise is 2-4.5e-4 on such patches
    (measured on BASELINE config 3's patches, profiles/r6/heavy_family_calibration.txt) -- a net whose reference output is only defined to
    half the tolerance tests nothing about this path.  Every class stays in use once the head bias is calibrated (calibrate_head_bias).
    """
    d, c, k = VIT_CONFIGS[name]
    sd = make_vit_state_dict(name, seed, depth=depth, head_gain=head_gain)
    tag = name + "/heavy/"

    def student_t3(key_name, shape):
        n = int(math.prod(shape))
        num = approx_normal(stream_key(seed, tag + key_name + "/n"), n)
        den = torch.zeros(n, dtype=torch.float64)
        for j in range(3):
            den += approx_normal(stream_key(seed, tag + key_name + f"/d{j}"), n) ** 2
        t = num / torch.sqrt(den / 3.0).clamp_min(1e-3)
        return (t / math.sqrt(3.0)).reshape(shape)          # variance of t(3) is 3

    def heavy_linear(key_name, out_f, in_f):
        std = lin_gain * math.sqrt(2.0 / (in_f + out_f))
        return (student_t3(key_name, (out_f, in_f)) * std).to(torch.float32)

    def gamma(key_name):
        u = uniform(stream_key(seed, tag + key_name), d)
        return torch.exp((2.0 * u - 1.0) * math.log(gamma_max)).to(torch.float32)

    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = gamma(p + "norm1.weight")
        sd[p + "norm2.weight"] = gamma(p + "norm2.weight")
        sd[p + "attn.qkv.weight"] = heavy_linear(p + "attn.qkv.weight", 3 * d, d)
        sd[p + "attn.proj.weight"] = heavy_linear(p + "attn.proj.weight", d, d)
        sd[p + "mlp.fc1.weight"] = heavy_linear(p + "mlp.fc1.weight", 4 * d, d)
        sd[p + "mlp.fc2.weight"] = heavy_linear(p + "mlp.fc2.weight", d, 4 * d)
    sd["norm.weight"] = gamma("norm.weight")
    # the massive-activation channels: distinct, spread over the width
    picks = hash_u24(stream_key(seed, tag + "outliers"), torch.arange(outlier_channels, dtype=torch.int64))
    chans = sorted({int((int(v) + 37 * j) % d) for j, v in enumerate(picks.tolist())})
    for ch in chans:
        sd["pos_embed"][..., ch] *= outlier_gain
        sd["cls_token"][..., ch] *= outlier_gain
    return sd


WEIGHT_FAMILIES = {"uniform": make_vit_state_dict, "heavy": make_vit_state_dict_heavy}


#: imputer panels: name -> number of channel tokens L (reference markerImputer.py:260-274)
MAE_PANELS: Dict[str, int] = {"immune_full": 15, "immune_extended": 10, "immune_base": 7}


def make_mae_state_dict(panel: str, seed: int, enc_depth: int = 12, dec_depth: int = 8, lin_gain: float = 1.5) -> Dict[str, torch.Tensor]:
    """Seeded fp32 state dict with the keys of the reference's ``MaskedAutoencoderViT`` checkpoints
    (markerImputer.py:69-110): encoder 768/12 heads over 1600-pixel channel tokens, decoder 512/8 heads, pred 512->1600."""
    L = MAE_PANELS[panel]
    sd: Dict[str, torch.Tensor] = {}
    tag = "mae/" + panel + "/"

    def nrm(key_name, shape, std):
        n = int(math.prod(shape))
        return (approx_normal(stream_key(seed, tag + key_name), n) * std).to(torch.float32).reshape(shape)

    def uni(key_name, shape, lo, hi):
        n = int(math.prod(shape))
        return (uniform(stream_key(seed, tag + key_name), n) * (hi - lo) + lo).to(torch.float32).reshape(shape)

    def xavier(key_name, out_f, in_f, gain=lin_gain):
        a = gain * math.sqrt(6.0 / (in_f + out_f))
        return uni(key_name, (out_f, in_f), -a, a)

    def blocks(prefix, depth, d):
        for i in range(depth):
            p = f"{prefix}{i}."
            sd[p + "norm1.weight"] = uni(p + "norm1.weight", (d,), 0.9, 1.1)
            sd[p + "norm1.bias"] = uni(p + "norm1.bias", (d,), -0.05, 0.05)
            sd[p + "attn.qkv.weight"] = xavier(p + "attn.qkv.weight", 3 * d, d)
            sd[p + "attn.qkv.bias"] = uni(p + "attn.qkv.bias", (3 * d,), -0.02, 0.02)
            sd[p + "attn.proj.weight"] = xavier(p + "attn.proj.weight", d, d)
            sd[p + "attn.proj.bias"] = uni(p + "attn.proj.bias", (d,), -0.02, 0.02)
            sd[p + "norm2.weight"] = uni(p + "norm2.weight", (d,), 0.9, 1.1)
            sd[p + "norm2.bias"] = uni(p + "norm2.bias", (d,), -0.05, 0.05)
            sd[p + "mlp.fc1.weight"] = xavier(p + "mlp.fc1.weight", 4 * d, d)
            sd[p + "mlp.fc1.bias"] = uni(p + "mlp.fc1.bias", (4 * d,), -0.02, 0.02)
            sd[p + "mlp.fc2.weight"] = xavier(p + "mlp.fc2.weight", d, 4 * d)
            sd[p + "mlp.fc2.bias"] = uni(p + "mlp.fc2.bias", (d,), -0.02, 0.02)

    sd["cls_token"] = nrm("cls_token", (1, 1, 768), 0.02)
    sd["pos_embed"] = uni("pos_embed", (1, L + 1, 768), -1.0, 1.0)            # sin-cos tables in the real checkpoints: O(1) values
    sd["patch_embed.proj.weight"] = xavier("patch_embed.proj.weight", 768, 1600, 2.0).reshape(768, 1, 40, 40)
    sd["patch_embed.proj.bias"] = uni("patch_embed.proj.bias", (768,), -0.02, 0.02)
    blocks("blocks.", enc_depth, 768)
    sd["norm.weight"] = uni("norm.weight", (768,), 0.9, 1.1)
    sd["norm.bias"] = uni("norm.bias", (768,), -0.05, 0.05)
    sd["decoder_embed.weight"] = xavier("decoder_embed.weight", 512, 768)
    sd["decoder_embed.bias"] = uni("decoder_embed.bias", (512,), -0.02, 0.02)
    sd["mask_token"] = nrm("mask_token", (1, 1, 512), 0.02)
    sd["decoder_pos_embed"] = uni("decoder_pos_embed", (1, L + 1, 512), -1.0, 1.0)
    blocks("decoder_blocks.", dec_depth, 512)
    sd["decoder_norm.weight"] = uni("decoder_norm.weight", (512,), 0.9, 1.1)
    sd["decoder_norm.bias"] = uni("decoder_norm.bias", (512,), -0.05, 0.05)
    sd["decoder_pred.weight"] = xavier("decoder_pred.weight", 1600, 512, 1.0)
    sd["decoder_pred.bias"] = uni("decoder_pred.bias", (1600,), -0.3, 0.3)
    return sd


def calibrate_head_bias(sd: Dict[str, torch.Tensor], features: torch.Tensor) -> torch.Tensor:
    """Head bias that centres the logits of a calibration batch (``features`` = LN(z)[:, 0], (n, D)):
    removes the cell-independent logit offset a random ViT has, so labels spread over several classes.
    Test/golden helper -- the calibrated bias is stored with the fixture, never recomputed on another box."""
    return -(features.to(torch.float64).mean(0) @ sd["head.weight"].to(torch.float64).t()).to(torch.float32)


# ----------------------------------------------------------------------------------------------
# segmentation mask + multiplexed image
# ----------------------------------------------------------------------------------------------

def _cell_table(h: int, w: int, n_cells: int, seed: int, device):
    """Jittered grid of discs. Returns pitch, grid dims and per-grid-cell (label, cy, cx, r2) tensors
    in 1/16 px fixed point (r2 in 1/256 px^2).  label 0 marks an unpopulated grid cell."""
    p = int(math.isqrt((h * w) // max(n_cells, 1)))
    p = max(p, 6)
    gy, gx = (h + p - 1) // p, (w + p - 1) // p
    g = gy * gx
    if n_cells > g:
        raise ValueError(f"cannot place {n_cells} cells on a {gy}x{gx} grid")
    idx = torch.arange(g, dtype=torch.int64, device=device)
    # which grid cells are populated, and with which label: two hashed permutations
    order = torch.argsort(hash_u24(stream_key(seed, "mask/populate"), idx) * g + idx)  # unique keys
    populated = order[:n_cells]
    lab_order = torch.argsort(hash_u24(stream_key(seed, "mask/label"), torch.arange(n_cells, dtype=torch.int64, device=device)) * n_cells
                              + torch.arange(n_cells, dtype=torch.int64, device=device))
    label = torch.zeros(g, dtype=torch.int64, device=device)
    label[populated] = lab_order + 1
    # centre jitter +-0.15 p, radius U(0.30 p, 0.48 p), all in 1/16 px
    p16 = p * 16
    jy = (hash_u24(stream_key(seed, "mask/jy"), idx) * (2 * 15 * p16 // 100 + 1) >> 24) - 15 * p16 // 100
    jx = (hash_u24(stream_key(seed, "mask/jx"), idx) * (2 * 15 * p16 // 100 + 1) >> 24) - 15 * p16 // 100
    cy = (idx // gx) * p16 + p16 // 2 + jy
    cx = (idx % gx) * p16 + p16 // 2 + jx
    r = 30 * p16 // 100 + (hash_u24(stream_key(seed, "mask/r"), idx) * (18 * p16 // 100 + 1) >> 24)
    return p, gy, gx, label, cy, cx, r * r


def make_mask_and_image(h: int, w: int, n_cells: int, n_channels: int, seed: int, device="cpu",
                        positive_frac: float = 0.3, want_image: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Synthetic segmentation mask ``(H, W) int32`` and raw image ``(C, H, W) uint16`` (as int32 tensor
    holding values 0..65535; callers cast).  Cells touch all four borders (grid starts at 0)."""
    p, gy, gx, label, cy, cx, r2 = _cell_table(h, w, n_cells, seed, device)
    g = gy * gx
    ys = torch.arange(h, dtype=torch.int64, device=device)
    xs = torch.arange(w, dtype=torch.int64, device=device)
    py = (ys * 16 + 8)[:, None]  # pixel centres, 1/16 px
    px = (xs * 16 + 8)[None, :]
    gyi = (ys // p)[:, None]
    gxi = (xs // p)[None, :]
    mask = torch.zeros((h, w), dtype=torch.int64, device=device)

    if want_image:
        # per (grid cell, channel): positive with prob positive_frac, peak log-uniform-ish in [200, 8000]
        cc = torch.arange(g * n_channels, dtype=torch.int64, device=device)
        pos = hash_u24(stream_key(seed, "img/pos"), cc) < int(positive_frac * (1 << 24))
        e = hash_u24(stream_key(seed, "img/peak"), cc)  # 24 bits
        # piecewise "log-uniform": 200 * 2^(u*5.32) approximated by octave + linear mantissa
        octv = (e >> 21) % 6  # 0..5  (top 3 bits, folded)
        mant = e & ((1 << 21) - 1)
        peak = (200 << octv) + (((200 << octv) * mant) >> 21)
        peak = torch.where(pos, peak, torch.zeros_like(peak)).reshape(g, n_channels)
        peak = torch.where((label > 0)[:, None], peak, torch.zeros_like(peak))
        acc = torch.zeros((n_channels, h, w), dtype=torch.int64, device=device)

    for dy in (0, -1, 1):
        for dx in (0, -1, 1):
            ny = gyi + dy
            nx = gxi + dx
            ok = (ny >= 0) & (ny < gy) & (nx >= 0) & (nx < gx)
            gi = (ny.clamp(0, gy - 1) * gx + nx.clamp(0, gx - 1))
            d2 = (py - cy[gi]) ** 2 + (px - cx[gi]) ** 2
            lab = label[gi]
            inside = ok & (lab > 0) & (d2 <= r2[gi])
            mask = torch.where((mask == 0) & inside, lab, mask)
            if want_image:
                # soft bump with support radius 1.3 r:  w = (1 - d2/R2)^2 in 12-bit fixed point
                big = (r2[gi] * 169) // 100
                t = ((big - d2).clamp(min=0) << 12) // big.clamp(min=1)
                wgt = (t * t) >> 12
                wgt = torch.where(ok & (lab > 0), wgt, torch.zeros_like(wgt))
                for c in range(n_channels):
                    acc[c] += (peak[:, c][gi] * wgt) >> 12

    if not want_image:
        return mask.to(torch.int32), None

    # smooth background: bilinear interpolation of a coarse hashed lattice (64 px pitch), 50..350 counts
    pitch = 64
    ly, lx = h // pitch + 2, w // pitch + 2
    fy = (ys % pitch)[:, None]
    fx = (xs % pitch)[None, :]
    iy = (ys // pitch)[:, None]
    ix = (xs // pitch)[None, :]
    for c in range(n_channels):
        lat = 50 + (hash_u24(stream_key(seed, "img/bg", c), torch.arange(ly * lx, dtype=torch.int64, device=device)) * 301 >> 24)
        lat = lat.reshape(ly, lx)
        v00 = lat[iy, ix]
        v01 = lat[iy, ix + 1]
        v10 = lat[iy + 1, ix]
        v11 = lat[iy + 1, ix + 1]
        bg = (v00 * (pitch - fy) * (pitch - fx) + v01 * (pitch - fy) * fx + v10 * fy * (pitch - fx) + v11 * fy * fx) // (pitch * pitch)
        pix = (ys[:, None] * w + xs[None, :]) + c * h * w
        noise = hash_u24(stream_key(seed, "img/noise"), pix) * 31 >> 24
        acc[c] += bg + noise
    return mask.to(torch.int32), acc.clamp(0, 65535).to(torch.int32)


def make_image_for_mask(mask: torch.Tensor, n_channels: int, seed: int, positive_frac: float = 0.3) -> torch.Tensor:
    """Raw image ``(C, H, W)`` (int32 tensor holding uint16 values) for a GIVEN label mask (e.g. the reference's
    ``examples/example_1_cell_mask.png``, BASELINE config 1 stand-in): every (label, channel) is positive with probability
    ``positive_frac`` with a flat in-cell intensity drawn like ``make_mask_and_image``'s peaks (200 .. 12800), on the same kind of
    smooth hashed background + noise.  Counter-based hashes only: identical on every box."""
    device = mask.device
    h, w = mask.shape
    top = int(mask.max()) + 1
    cc = torch.arange(top * n_channels, dtype=torch.int64, device=device)
    pos = hash_u24(stream_key(seed, "gimg/pos"), cc) < int(positive_frac * (1 << 24))
    e = hash_u24(stream_key(seed, "gimg/peak"), cc)
    octv = (e >> 21) % 6
    mant = e & ((1 << 21) - 1)
    peak = (200 << octv) + (((200 << octv) * mant) >> 21)
    peak = torch.where(pos, peak, torch.zeros_like(peak)).reshape(top, n_channels)
    peak[0] = 0
    acc = peak[mask.to(torch.int64)].permute(2, 0, 1).contiguous()
    ys = torch.arange(h, dtype=torch.int64, device=device)
    xs = torch.arange(w, dtype=torch.int64, device=device)
    pitch = 64
    ly, lx = h // pitch + 2, w // pitch + 2
    fy, fx = (ys % pitch)[:, None], (xs % pitch)[None, :]
    iy, ix = (ys // pitch)[:, None], (xs // pitch)[None, :]
    for c in range(n_channels):
        lat = 50 + (hash_u24(stream_key(seed, "gimg/bg", c), torch.arange(ly * lx, dtype=torch.int64, device=device)) * 301 >> 24)
        lat = lat.reshape(ly, lx)
        bg = (lat[iy, ix] * (pitch - fy) * (pitch - fx) + lat[iy, ix + 1] * (pitch - fy) * fx + lat[iy + 1, ix] * fy * (pitch - fx)
              + lat[iy + 1, ix + 1] * fy * fx) // (pitch * pitch)
        pix = (ys[:, None] * w + xs[None, :]) + c * h * w
        acc[c] += bg + (hash_u24(stream_key(seed, "gimg/noise"), pix) * 31 >> 24)
    return acc.clamp(0, 65535).to(torch.int32)


FULL_PANEL_MARKERS: List[str] = ['DAPI', 'CD3', 'CD4', 'CD8', 'CD11c', 'CD15', 'CD20', 'CD45', 'CD56', 'CD68', 'CD138',
                                 'CD163', 'FoxP3', 'Granzyme B', 'Trypase']
BASIC_PANEL_MARKERS: List[str] = ['CD45', 'CD20', 'CD4', 'CD8', 'DAPI', 'CD11c', 'CD3']
