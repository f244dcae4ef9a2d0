"""Emulation study (CPU, oracle ViTs): LayerNorm folded into the following GEMM, residual stream kept in the packed-split format.
   y = LN(z) W^T + b  ==  rstd * ( z (gamma o W)^T - mu * c ) + (beta W^T + b),   c_n = sum_k (gamma o W)_nk
   every product fp16 hi+lo x3 (as production); z re-quantised to hi+lo after every residual update."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import precision_study as ps
from oracle import ref_vit
from multiplexed_image_annotator_amd import synth

def q22(x):
    h = ps.r_f16(x); return h + ps.r_f16(x - h)

def mm3(a, b):   # fp16x3 product, fp32 accumulate (a (..,m,k), b (n,k))
    ah = ps.r_f16(a); al = ps.r_f16(a - ah); bh = ps.r_f16(b); bl = ps.r_f16(b - bh)
    return torch.cat((ah, al, ah), -1) @ torch.cat((bh, bh, bl), -1).transpose(-2, -1)

@torch.no_grad()
def logits_folded(sd, x, fold=True, quant_z=True):
    heads = ref_vit.HEADS; b = x.shape[0]; d = sd["cls_token"].shape[-1]; hd = d // heads
    t = ref_vit.patch_embed(sd, x)
    z = torch.cat((sd["cls_token"].expand(b, -1, -1), t), dim=1) + sd["pos_embed"]
    if quant_z: z = q22(z)
    n = z.shape[1]
    def ln_gemm(z, gw, gb, w, bias):
        if not fold:
            return mm3(F.layer_norm(z, (d,), gw, gb, ref_vit.LN_EPS), w) + bias
        wp = (w.double() * gw.double()).float()                 # gamma o W, rounded to fp32 then split like any weight
        wph = ps.r_f16(wp); wpl = ps.r_f16(wp - wph)
        c = (wph.double() + wpl.double()).sum(1).float()        # c_n from the SAME rounded operand
        bp = (w.double() @ gb.double() + bias.double()).float()
        mu = z.mean(-1, keepdim=True)
        var = ((z - mu) ** 2).mean(-1, keepdim=True)
        rstd = torch.rsqrt(var + ref_vit.LN_EPS)
        acc = mm3(z, wp)
        return rstd * (acc - mu * c) + bp
    for i in range(ref_vit.depth_of(sd)):
        p = f"blocks.{i}."
        qkv = ln_gemm(z, sd[p + "norm1.weight"], sd[p + "norm1.bias"], sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(b, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
        att = mm3(q, k).softmax(dim=-1)
        y = mm3(att, v.transpose(-2, -1)).transpose(1, 2).reshape(b, n, d)
        z = z + mm3(y, sd[p + "attn.proj.weight"]) + sd[p + "attn.proj.bias"]
        if quant_z: z = q22(z)
        h = F.gelu(ln_gemm(z, sd[p + "norm2.weight"], sd[p + "norm2.bias"], sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        z = z + mm3(h, sd[p + "mlp.fc2.weight"]) + sd[p + "mlp.fc2.bias"]
        if quant_z: z = q22(z)
    z = F.layer_norm(z, (d,), sd["norm.weight"], sd["norm.bias"], ref_vit.LN_EPS)
    return z[:, 0] @ sd["head.weight"].t() + sd["head.bias"]

torch.set_num_threads(8)
cells, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 64
seed = synth.SEED_BASE + 7
for name in (sys.argv[2:] or ["nerve", "immune_base", "struct", "immune_extended", "immune_full"]):
    sd = synth.make_vit_state_dict(name, seed)
    x = ps.patch_like_inputs(name, cells, seed + 1)
    ref = torch.cat([F.softmax(ref_vit.logits(sd, x[i:i + batch]), dim=1) for i in range(0, cells, batch)])
    for label, kw in (("x3 plain", dict(fold=False, quant_z=False)), ("x3 + PS residual stream", dict(fold=False, quant_z=True)),
                      ("x3 + PS stream + folded LN", dict(fold=True, quant_z=True))):
        t0 = time.time()
        got = torch.cat([F.softmax(logits_folded(sd, x[i:i + batch], **kw), dim=1) for i in range(0, cells, batch)])
        dp = (got - ref).abs()
        print(f"{name:16s} {label:28s} max|dp| {dp.max():.2e} mean {dp.mean():.2e} flips {int((got.argmax(1) != ref.argmax(1)).sum())} ({time.time()-t0:.0f} s)", flush=True)
