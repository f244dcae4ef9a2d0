"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement, in plain fp32 torch, of the ViT classifier forward the reference reaches through
``timm`` (timm <= 1.0.14 is a dependency that is NOT vendored under /root/reference and is not
installed here; its published ``VisionTransformer``/``PatchEmbed``/``Block``/``Attention``/``Mlp``
semantics are restated).  Parity anchors: the reference's own call sites

* ``cell_type_annotation/model.py:31-64``  -- subclass ``forward_features`` (patch_embed, CLS concat,
  ``+ pos_embed``, blocks, ``norm``, token 0) with ``global_pool=False``;
* ``cell_type_annotation/model.py:66-88``  -- factories: patch 4, depth 12, 12 heads, mlp_ratio 4,
  qkv_bias, ``LayerNorm(eps=1e-6)``;
* ``cell_type_annotation/model.py:188-234`` -- per-model (embed_dim, in_chans, num_classes), img_size 40;
* ``cell_type_annotation/model.py:401-404`` -- ``softmax(model(x), dim=1)``.

The restatement is cross-checked in ``tests/test_oracle_vit.py`` against an independent implementation
of the same pre-LN ViT (``transformers.ViTForImageClassification``), because no reference fixture
pins it ("parity unpinned" for third-party arithmetic).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

PATCH = 4
IMG = 40
HEADS = 12
LN_EPS = 1e-6


def depth_of(sd: Dict[str, torch.Tensor]) -> int:
    n = 0
    while f"blocks.{n}.norm1.weight" in sd:
        n += 1
    return n


def patch_embed(sd, x: torch.Tensor) -> torch.Tensor:
    """timm PatchEmbed: Conv2d(C, D, k=4, s=4) -> flatten(2).transpose(1, 2): tokens row-major (py*10+px)."""
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=PATCH)
    return t.flatten(2).transpose(1, 2)


def block(sd, i: int, z: torch.Tensor, heads: int = HEADS) -> torch.Tensor:
    """timm Block (eval: DropPath / Dropout / LayerScale are identities)."""
    p = f"blocks.{i}."
    b, n, d = z.shape
    hd = d // heads
    y = F.layer_norm(z, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
    qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(b, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q * hd ** -0.5) @ k.transpose(-2, -1)
    att = att.softmax(dim=-1)
    y = (att @ v).transpose(1, 2).reshape(b, n, d)
    z = z + F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    y = F.layer_norm(z, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
    y = F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    y = F.gelu(y)  # exact erf GELU (nn.GELU default)
    z = z + F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return z


def forward_features(sd, x: torch.Tensor) -> torch.Tensor:
    """reference model.py:45-64 with global_pool=False: returns LN(z)[:, 0]."""
    b = x.shape[0]
    t = patch_embed(sd, x)
    z = torch.cat((sd["cls_token"].expand(b, -1, -1), t), dim=1) + sd["pos_embed"]
    for i in range(depth_of(sd)):
        z = block(sd, i, z)
    d = z.shape[-1]
    z = F.layer_norm(z, (d,), sd["norm.weight"], sd["norm.bias"], LN_EPS)
    return z[:, 0]


def logits(sd, x: torch.Tensor) -> torch.Tensor:
    """timm forward_head with the subclass's ``global_pool=False`` (pool skipped, fc_norm Identity, head)."""
    return F.linear(forward_features(sd, x), sd["head.weight"], sd["head.bias"])


@torch.no_grad()
def predict_proba(sd, x: torch.Tensor, batch_size: int = 128) -> torch.Tensor:
    """reference model.py:397-406: sub-batched forward + softmax(dim=1), fp32."""
    out = []
    for i in range(0, x.shape[0], batch_size):
        out.append(F.softmax(logits(sd, x[i:i + batch_size].to(torch.float32)), dim=1))
    if not out:
        k = sd["head.weight"].shape[0]
        return torch.zeros((0, k), dtype=torch.float32)
    return torch.cat(out, dim=0)


# ----------------------------------------------------------------------------------------------
# precision study helper (used by tests/tools only): emulate operand rounding of a matrix-core path
# ----------------------------------------------------------------------------------------------

@torch.no_grad()
def logits_emulated(sd, x: torch.Tensor, operand_dtype: Optional[torch.dtype]) -> torch.Tensor:
    """Same forward with every matmul operand rounded to ``operand_dtype`` (fp32 accumulate), the
    residual stream, LayerNorm statistics and softmax kept in fp32 -- the numerics of the HIP path."""
    def q(t):
        return t.to(operand_dtype).to(torch.float32) if operand_dtype is not None else t

    b = x.shape[0]
    d = sd["cls_token"].shape[-1]
    hd = d // HEADS
    c = x.shape[1]
    cols = x.reshape(b, c, 10, 4, 10, 4).permute(0, 2, 4, 1, 3, 5).reshape(b, 100, c * 16)
    w = sd["patch_embed.proj.weight"].reshape(d, -1)
    t = q(cols) @ q(w).t() + sd["patch_embed.proj.bias"]
    z = torch.cat((sd["cls_token"].expand(b, -1, -1), t), dim=1) + sd["pos_embed"]
    for i in range(depth_of(sd)):
        p = f"blocks.{i}."
        y = F.layer_norm(z, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
        qkv = q(q(y) @ q(sd[p + "attn.qkv.weight"]).t() + sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(b, -1, 3, HEADS, hd).permute(2, 0, 3, 1, 4)
        qq, kk, vv = qkv[0], qkv[1], qkv[2]
        att = (qq @ kk.transpose(-2, -1)) * hd ** -0.5
        att = att.softmax(dim=-1)
        y = q(q(att) @ vv).transpose(1, 2).reshape(b, -1, d)
        z = z + y @ q(sd[p + "attn.proj.weight"]).t() + sd[p + "attn.proj.bias"]
        y = F.layer_norm(z, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
        y = q(y) @ q(sd[p + "mlp.fc1.weight"]).t() + sd[p + "mlp.fc1.bias"]
        y = q(F.gelu(y))
        z = z + y @ q(sd[p + "mlp.fc2.weight"]).t() + sd[p + "mlp.fc2.bias"]
    z = F.layer_norm(z, (d,), sd["norm.weight"], sd["norm.bias"], LN_EPS)
    return z[:, 0] @ sd["head.weight"].t() + sd["head.bias"]
