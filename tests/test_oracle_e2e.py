"""Oracle end-to-end pipeline vs the reference's own Annotator (preprocess -> predict -> export_annotations) goldens."""
import hashlib
import json
import os

import numpy as np
import pytest

from multiplexed_image_annotator_amd import synth
from oracle import ref_pipeline


def load_case(golden_dir, name, tmp_path):
    meta = json.load(open(os.path.join(golden_dir, "e2e.json")))[name]
    arrs = np.load(os.path.join(golden_dir, "e2e.npz"))
    mask, img = synth.make_mask_and_image(meta["h"], meta["w"], meta["cells"], len(meta["markers"]), meta["seed"])
    raw = img.numpy().astype(np.uint16)
    mask = mask.numpy().astype(np.int32)
    assert hashlib.sha256(raw.tobytes()).hexdigest() == meta["img_sha"]
    assert hashlib.sha256(mask.tobytes()).hexdigest() == meta["mask_sha"]
    weights = {}
    for m in meta["models"]:
        sd = synth.make_vit_state_dict(m, meta["seed"])
        import torch
        sd["head.bias"] = torch.from_numpy(arrs[f"{name}__head_bias_{m}"])
        weights[m] = sd
    mf = tmp_path / "markers.txt"
    mf.write_text("\n".join(meta["markers"]) + "\n")
    return meta, arrs, raw, mask, weights, str(mf)


@pytest.mark.parametrize("name", ["basic", "two_model"])
def test_e2e_matches_reference(golden_dir, tmp_path, name):
    meta, arrs, raw, mask, weights, mf = load_case(golden_dir, name, tmp_path)
    r = ref_pipeline.run_image(raw, mask, mf, weights, strict=meta["strict"], normalize=True, blur=meta["blur"], amax=meta["amax"],
                               confidence=meta["conf"], batch_size=8)
    assert r["ids"].tolist() == meta["cell_ids"]
    for m in meta["models"]:
        np.testing.assert_allclose(r["probs"][m], arrs[f"{name}__p_{m}"], rtol=0, atol=5e-6)
    assert r["labels"] == meta["labels"]
    assert [str(s) for s in r["cell_types"]] == meta["cell_types"]
    assert r["type_ints"] == meta["type_ints"]
    np.testing.assert_allclose(np.array([np.float32(c) for c in r["conf"]]), arrs[f"{name}__conf"], rtol=0, atol=5e-6)
    np.testing.assert_array_equal(r["intensity"], arrs[f"{name}__intensity"])
    # CSV: identical text except confidences, which may differ in the last fp32 digits between two CPU fp32 runs
    got = r["csv"].splitlines()
    exp = meta["csv"].splitlines()
    assert len(got) == len(exp) and got[0] == exp[0]
    for a, b in zip(got[1:], exp[1:]):
        fa, fb = a.split(","), b.split(",")
        assert fa[:2] == fb[:2] and fa[3:] == fb[3:]
        assert abs(float(fa[2]) - float(fb[2])) <= 1.5e-3


def load_config1(golden_dir, tmp_path):
    """BASELINE.json configs[0] stand-in: the reference's example_1 mask (1850 cells, from cellpos.npz) + the seeded 7-channel image."""
    import torch
    meta = json.load(open(os.path.join(golden_dir, "config1.json")))
    arrs = np.load(os.path.join(golden_dir, "config1.npz"))
    mask = np.load(os.path.join(golden_dir, "cellpos.npz"))["example1_mask"].astype(np.int32)
    raw = synth.make_image_for_mask(torch.from_numpy(mask), len(meta["markers"]), meta["seed"]).numpy().astype(np.uint16)
    assert hashlib.sha256(raw.tobytes()).hexdigest() == meta["img_sha"]
    assert hashlib.sha256(mask.tobytes()).hexdigest() == meta["mask_sha"]
    sd = synth.make_vit_state_dict("immune_base", meta["seed"])
    sd["head.bias"] = torch.from_numpy(arrs["head_bias"])
    mf = tmp_path / "markers.txt"
    mf.write_text("\n".join(meta["markers"]) + "\n")
    return meta, arrs, raw, mask, {"immune_base": sd}, str(mf)


def test_config1_matches_reference(golden_dir, tmp_path):
    """The oracle pipeline on BASELINE config 1's stand-in against the reference Annotator's own run of it (CPU, bs = 8)."""
    import torch
    torch.set_num_threads(min(8, len(os.sched_getaffinity(0))))
    meta, arrs, raw, mask, weights, mf = load_config1(golden_dir, tmp_path)
    r = ref_pipeline.run_image(raw, mask, mf, weights, strict=True, normalize=True, blur=meta["blur"], amax=meta["amax"],
                               confidence=meta["conf"], batch_size=meta["batch_size"])
    assert len(r["ids"]) == meta["cells"] == 1850
    np.testing.assert_allclose(r["probs"]["immune_base"], arrs["probs"], rtol=0, atol=5e-6)
    assert r["labels"] == meta["labels"]
    assert [str(s) for s in r["cell_types"]] == meta["cell_types"]
    np.testing.assert_array_equal(r["intensity"], arrs["intensity"])
    got, exp = r["csv"].splitlines(), meta["csv"].splitlines()
    assert len(got) == len(exp) and got[0] == exp[0]
    for a, b in zip(got[1:], exp[1:]):
        fa, fb = a.split(","), b.split(",")
        assert fa[:2] == fb[:2] and fa[3:] == fb[3:]
        assert abs(float(fa[2]) - float(fb[2])) <= 1.5e-3
