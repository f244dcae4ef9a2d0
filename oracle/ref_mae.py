"""ORACLE (test infrastructure).  Restatement of the reference's marker imputer
(``cell_type_annotation/markerImputer.py:69-329``): a masked auto-encoder whose tokens are the panel's CHANNELS
(each 40x40 channel patch is one token of 1600 pixels): encoder ViT-B (768, 12 blocks, 12 heads) over the present
channels + CLS, decoder (512, 8 blocks, 8 heads) over all L channels + CLS, linear prediction of the 1600 pixels of every
masked (= missing) channel, blended back into the patch tensor.

timm's ``PatchEmbed`` / ``Block`` semantics are restated as in ``ref_vit`` (timm is not installed, "parity unpinned" for
that third-party arithmetic); the reference's own bookkeeping (mosaic order, noise / argsort masking, ids_restore,
unpatchify, blend, write-back) is pinned by tests/golden/mae.npz, produced by the reference's ``MarkerImputer.impute``.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F

#: panel -> (channels L, mosaic grid h x w)   markerImputer.py:260-274
PANEL_GRID = {"immune_full": (15, (3, 5)), "immune_extended": (10, (2, 5)), "immune_base": (7, (1, 7))}
ENC_DIM, ENC_HEADS, DEC_DIM, DEC_HEADS = 768, 12, 512, 8
LN_EPS = 1e-6


def _block(sd, prefix: str, z: torch.Tensor, heads: int) -> torch.Tensor:
    b, n, d = z.shape
    hd = d // heads
    y = F.layer_norm(z, (d,), sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"], LN_EPS)
    qkv = F.linear(y, sd[prefix + "attn.qkv.weight"], sd[prefix + "attn.qkv.bias"]).reshape(b, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = ((qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1)).softmax(dim=-1)
    y = (att @ qkv[2]).transpose(1, 2).reshape(b, n, d)
    z = z + F.linear(y, sd[prefix + "attn.proj.weight"], sd[prefix + "attn.proj.bias"])
    y = F.layer_norm(z, (d,), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], LN_EPS)
    y = F.gelu(F.linear(y, sd[prefix + "mlp.fc1.weight"], sd[prefix + "mlp.fc1.bias"]))
    return z + F.linear(y, sd[prefix + "mlp.fc2.weight"], sd[prefix + "mlp.fc2.bias"])


def _depth(sd, prefix: str) -> int:
    n = 0
    while f"{prefix}{n}.norm1.weight" in sd:
        n += 1
    return n


@torch.no_grad()
def impute(sd: Dict[str, torch.Tensor], data: torch.Tensor, present: Sequence[int], batch_size: int = 64) -> torch.Tensor:
    """markerImputer.py:294-329.  ``data`` (n, L, 40, 40) fp32; ``present`` = positions whose marker exists
    (``channel_index`` of the reference).  Returns data with every other channel replaced by the MAE prediction."""
    n, L = data.shape[0], data.shape[1]
    present = list(present)
    keep = torch.tensor(present, dtype=torch.long)
    missing = [c for c in range(L) if c not in present]
    out = data.clone()
    w = sd["patch_embed.proj.weight"].reshape(ENC_DIM, 1600)
    for s in range(0, n, batch_size):
        x = data[s:s + batch_size]
        b = x.shape[0]
        # forward_encoder (:186-206): embed every channel tile, add pos (w/o cls), keep the present ones, prepend cls
        tok = F.linear(x.reshape(b, L, 1600), w, sd["patch_embed.proj.bias"]) + sd["pos_embed"][:, 1:, :]
        z = torch.cat(((sd["cls_token"] + sd["pos_embed"][:, :1, :]).expand(b, -1, -1), tok[:, keep]), dim=1)
        for i in range(_depth(sd, "blocks.")):
            z = _block(sd, f"blocks.{i}.", z, ENC_HEADS)
        z = F.layer_norm(z, (ENC_DIM,), sd["norm.weight"], sd["norm.bias"], LN_EPS)
        # forward_decoder (:208-232): project, put mask tokens at the missing positions (ids_restore), add pos, decode
        y = F.linear(z, sd["decoder_embed.weight"], sd["decoder_embed.bias"])
        full = sd["mask_token"].expand(b, L, -1).clone()
        full[:, keep] = y[:, 1:]
        d = torch.cat((y[:, :1], full), dim=1) + sd["decoder_pos_embed"]
        for i in range(_depth(sd, "decoder_blocks.")):
            d = _block(sd, f"decoder_blocks.{i}.", d, DEC_HEADS)
        d = F.layer_norm(d, (DEC_DIM,), sd["decoder_norm.weight"], sd["decoder_norm.bias"], LN_EPS)
        pred = F.linear(d, sd["decoder_pred.weight"], sd["decoder_pred.bias"])[:, 1:]      # (b, L, 1600)
        for c in missing:                                                                   # blend (:316) + write-back (:324-326)
            out[s:s + b, c] = pred[:, c].reshape(b, 40, 40)
    return out
