"""Oracle ViT restatement: vs the reference's own VisionTransformer subclass run through the timm stand-in (golden),
and vs an independent implementation of the same pre-LN ViT (transformers.ViTForImageClassification)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from multiplexed_image_annotator_amd import synth
from oracle import ref_vit


def vit_inputs(model_name, n, seed):
    d, c, k = synth.VIT_CONFIGS[model_name]
    u = synth.uniform(synth.stream_key(seed, "vitx/" + model_name), n * c * 1600).reshape(n, c, 40, 40).to(torch.float32)
    x = u * 2 - 1
    return torch.where(x > 0.1, x, torch.full_like(x, -1.0))


@pytest.mark.parametrize("name", list(synth.VIT_CONFIGS))
def test_vit_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "vit_logits.npz"))
    x = vit_inputs(name, 8, synth.SEED_BASE + 7)
    assert hashlib.sha256(x.numpy().tobytes()).digest() == g[name + "_x_sha"].tobytes()  # generator is deterministic
    sd = synth.make_vit_state_dict(name, synth.SEED_BASE + 7)
    with torch.no_grad():
        lg = ref_vit.logits(sd, x)
    np.testing.assert_allclose(lg.numpy(), g[name + "_logits"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ref_vit.predict_proba(sd, x, 3).numpy(), g[name + "_probs"], rtol=0, atol=2e-6)


def test_vit_matches_hf_implementation():
    transformers = pytest.importorskip("transformers")
    name = "immune_base"
    d, c, k = synth.VIT_CONFIGS[name]
    sd = synth.make_vit_state_dict(name, 11, depth=3)
    cfg = transformers.ViTConfig(hidden_size=d, num_hidden_layers=3, num_attention_heads=12, intermediate_size=4 * d,
                                 hidden_act="gelu", layer_norm_eps=1e-6, image_size=40, patch_size=4, num_channels=c,
                                 qkv_bias=True, num_labels=k, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hf = transformers.ViTForImageClassification(cfg).eval()
    m = {"vit.embeddings.cls_token": sd["cls_token"], "vit.embeddings.position_embeddings": sd["pos_embed"],
         "vit.embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
         "vit.embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
         "vit.layernorm.weight": sd["norm.weight"], "vit.layernorm.bias": sd["norm.bias"],
         "classifier.weight": sd["head.weight"], "classifier.bias": sd["head.bias"]}
    have = set(hf.state_dict().keys())
    new_names = "vit.layers.0.attention.q_proj.weight" in have      # transformers >= 5 naming
    for i in range(3):
        p = f"blocks.{i}."
        q = f"vit.layers.{i}." if new_names else f"vit.encoder.layer.{i}."
        w, b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj") if new_names else ("query", "key", "value")):
            pre = q + (f"attention.{nm}." if new_names else f"attention.attention.{nm}.")
            m[pre + "weight"] = w[j * d:(j + 1) * d]
            m[pre + "bias"] = b[j * d:(j + 1) * d]
        o = q + ("attention.o_proj." if new_names else "attention.output.dense.")
        m[o + "weight"] = sd[p + "attn.proj.weight"]
        m[o + "bias"] = sd[p + "attn.proj.bias"]
        m[q + "layernorm_before.weight"] = sd[p + "norm1.weight"]
        m[q + "layernorm_before.bias"] = sd[p + "norm1.bias"]
        m[q + "layernorm_after.weight"] = sd[p + "norm2.weight"]
        m[q + "layernorm_after.bias"] = sd[p + "norm2.bias"]
        f1 = q + ("mlp.fc1." if new_names else "intermediate.dense.")
        f2 = q + ("mlp.fc2." if new_names else "output.dense.")
        m[f1 + "weight"] = sd[p + "mlp.fc1.weight"]
        m[f1 + "bias"] = sd[p + "mlp.fc1.bias"]
        m[f2 + "weight"] = sd[p + "mlp.fc2.weight"]
        m[f2 + "bias"] = sd[p + "mlp.fc2.bias"]
    missing, unexpected = hf.load_state_dict(m, strict=False)
    assert not unexpected and all("pooler" in k for k in missing), (missing, unexpected)
    x = vit_inputs(name, 4, 3)
    with torch.no_grad():
        ref = hf(pixel_values=x).logits
        got = ref_vit.logits(sd, x)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=5e-5)


def test_flops_formula():
    tot = sum(synth.vit_flops_per_cell(n) for n in synth.VIT_CONFIGS)
    assert abs(tot / 1e9 - 20.2453) < 1e-3
    assert abs(synth.vit_flops_per_cell("immune_full") / 1e9 - 9.9604) < 1e-3
