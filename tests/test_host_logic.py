"""Product host logic (no GPU needed): marker parsing against the reference goldens, channel resolution, sharding."""
import json
import os

import numpy as np
import pytest


def test_marker_parser_matches_reference(golden_dir, tmp_path):
    from multiplexed_image_annotator_amd.marker_parse import MarkerParser
    cases = json.load(open(os.path.join(golden_dir, "parser_cases.json")))
    for key, case in cases.items():
        name, mode = key.split("|")
        f = tmp_path / (name + ".txt")
        f.write_text("\n".join(case["markers_in"]) + "\n")
        p = MarkerParser(strict=(mode == "strict"))
        p.parse(str(f))
        got = {k: (None if v is None else [int(i) for i in v]) for k, v in p.indices.items()}
        assert got == case["indices"], key
        assert [p.immune_base, p.immune_extended, p.immune_full, p.struct, p.nerve] == case["flags"], key
        assert [str(m) for m in p.markers] == case["markers"], key
        assert list(p.panels) == ["immune_base", "immune_extended", "immune_full", "structure", "nerve_cell"]


def test_resolve_channels_quirk():
    from multiplexed_image_annotator_amd.ops import resolve_channels
    assert resolve_channels([0, 1, 2], 5) == [0, 1, 2]
    assert resolve_channels([0, -1, 2, -1, -1], 7) == [0, -1, 2, 6, 6]   # first blank, later ones alias the last channel


def test_gaussian_taps_match_scipy():
    from scipy.ndimage import gaussian_filter1d
    from multiplexed_image_annotator_amd.ops import gaussian_taps
    taps = gaussian_taps()
    off = 0
    for sigma in (1, 2, 3):
        r = 4 * sigma
        impulse = np.zeros(6 * r + 1)
        impulse[3 * r] = 1.0
        resp = gaussian_filter1d(impulse, sigma, mode="nearest", truncate=4.0)
        np.testing.assert_array_equal(resp[3 * r:3 * r + r + 1], taps[off:off + r + 1])
        off += r + 1


@pytest.mark.parametrize("ps", [20, 26, 39, 41, 45, 53, 60, 80, 90])
def test_resize_plan_matches_scipy_zoom(ps):
    """ops.resize_plan: the 40 nearest-neighbour source indices and the anti-alias taps are what scipy.ndimage.zoom(order=0,
    grid_mode=True) / gaussian_filter use (the two library calls inside skimage.transform.resize, reference preprocess.py:106)."""
    import numpy as np
    from scipy import ndimage as ndi
    from multiplexed_image_annotator_amd import ops
    taps, radius, idx = ops.resize_plan(ps)
    ramp = np.arange(ps * ps, dtype=np.float64).reshape(ps, ps)
    z = ndi.zoom(ramp, [40 / ps * 1.0 if False else 1.0 / (ps / 40)] * 2, order=0, mode="mirror", grid_mode=True)
    assert z.shape == (40, 40)
    np.testing.assert_array_equal(z, ramp[np.ix_(idx, idx)])
    sigma = max(0.0, (ps / 40 - 1) / 2)
    if sigma > 0 and int(4 * sigma + 0.5) > 0:
        imp = np.zeros(4 * radius + 1)
        imp[2 * radius] = 1.0
        ref = ndi.gaussian_filter(imp, sigma, mode="mirror")
        np.testing.assert_array_equal(ref[2 * radius:3 * radius + 1], taps)
    else:
        assert taps is None and radius == 0


def test_palette_and_confidence_colours_match_reference(golden_dir):
    """colors.get_colors / confidence_colors (host tables of the label painting) vs the reference's utils.get_colors / number_to_rgb."""
    import os
    import numpy as np
    from multiplexed_image_annotator_amd import colors
    g = np.load(os.path.join(golden_dir, "colorize.npz"))
    for n in (1, 2, 6, 17, 18, 19, 30):
        assert np.array_equal(np.array(colors.get_colors(n)), g[f"colors_{n}"])
    v = g["viridis_in"]
    got = colors.confidence_colors(v)
    exp = g["viridis_rgb"].copy()
    exp[~(v > 0)] = (192, 192, 192)                     # colorize paints non-positive confidences silver (model.py:831)
    assert np.array_equal(got.astype(np.int64), exp)


class _NotATensor:      # module-level so that pickle can name it (what an arbitrary-code checkpoint looks like to the loader)
    pass


def test_checkpoint_loader_executes_nothing_from_the_file(tmp_path):
    """annotator._load_state_dict: the reference's ``{"model": state_dict}`` layout (model.py:189-231) through torch's weights_only
    loader; a checkpoint holding anything but tensors is refused with a message naming the file, not unpickled."""
    import torch
    from multiplexed_image_annotator_amd.annotator import _load_state_dict
    good = tmp_path / "immune_base.pth"
    sd = {"cls_token": torch.arange(6, dtype=torch.float32).reshape(1, 1, 6), "head.bias": torch.zeros(3)}
    torch.save({"model": sd}, str(good))
    got = _load_state_dict(str(good))
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    bad = tmp_path / "evil.pth"
    torch.save({"model": {"w": torch.zeros(1)}, "extra": _NotATensor()}, str(bad))
    with pytest.raises(RuntimeError, match="evil.pth"):
        _load_state_dict(str(bad))
    nokey = tmp_path / "nokey.pth"
    torch.save({"state": sd}, str(nokey))
    with pytest.raises(RuntimeError, match="'model'"):
        _load_state_dict(str(nokey))
    # an MAE-style training checkpoint: args Namespace, epoch, optimizer state beside "model" (what the reference's torch.load of
    # model.py:189-231 accepts) loads through the allow-list, still without executing anything from the file
    import argparse
    mae = tmp_path / "mae_style.pth"
    torch.save({"model": sd, "args": argparse.Namespace(lr=1e-3, model="vit", blr=[1, 2]), "epoch": 7,
                "optimizer": {"state": {0: {"exp_avg": torch.zeros(2)}}, "param_groups": [{"lr": 1e-3, "params": [0]}]}}, str(mae))
    got = _load_state_dict(str(mae))
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    # I/O errors are not "re-save your checkpoint" errors
    with pytest.raises(FileNotFoundError):
        _load_state_dict(str(tmp_path / "missing.pth"))
    trunc = tmp_path / "truncated.pth"
    trunc.write_bytes(good.read_bytes()[:100])
    with pytest.raises(Exception) as ei:
        _load_state_dict(str(trunc))
    assert "Re-save it" not in str(ei.value)


def test_decision_distance_covers_every_comparison_of_the_vote():
    """ops.decision_distance: distance of a cell from the nearest boundary of merge_by_voting (reference model.py:481-633) -- top-2 margin of
    the candidate classes, the models' "Others" probabilities, the confidence and per-type thresholds"""
    import torch
    from multiplexed_image_annotator_amd.ops import decision_distance
    pa = torch.tensor([[0.50, 0.30, 0.20], [0.40, 0.399, 0.201], [0.26, 0.04, 0.70], [0.90, 0.05, 0.05]])      # Others = column 2
    pb = torch.tensor([[0.10, 0.90], [0.10, 0.90], [0.2505, 0.7495], [0.05, 0.95]])                              # Others = column 1
    d = decision_distance(pa, 2, pb, 1, [0.25])
    # cell 0: candidates .5 .3 | .1 -> margin .2, others .2 / .9, threshold .25 -> nearest: |.5 - .25| = .25? no: margin .2 and |.5 - .2| = .3 -> .2
    assert abs(d[0].item() - 0.2) < 1e-6
    assert abs(d[1].item() - 0.001) < 1e-6          # two candidates 1e-3 apart
    assert abs(d[2].item() - 0.0095) < 1e-6         # best candidate .26 against the other model's best .2505: margin .0095 (threshold .25 is .01 away)
    assert abs(d[3].item() - 0.05) < 1e-6           # .9 against "Others" .95 of the second model
    single = decision_distance(pa, 2, None, None, [0.3, -1])
    assert abs(single[0].item() - 0.2) < 1e-6 and abs(single[2].item() - 0.04) < 1e-6      # cell 2: |.26 - .3|


def test_effective_chunk_is_clamped():
    """the per-width chunk scale (x 4 at D = 576) never pushes a forward beyond MAX_SCALED_CHUNK cells; a caller that asks for
    more than that itself gets what it asked for, unscaled"""
    from multiplexed_image_annotator_amd import ops

    class M:      # the two things effective_chunk reads, without a GPU
        CHUNK_SCALE, MAX_SCALED_CHUNK = ops.VitModel.CHUNK_SCALE, ops.VitModel.MAX_SCALED_CHUNK
        chunk_scale = ops.VitModel.chunk_scale
        effective_chunk = ops.VitModel.effective_chunk

        def __init__(self, d):
            self.D = d

    assert M(576).effective_chunk(1024) == 4096 and M(576).effective_chunk(256) == 1024
    assert M(576).effective_chunk(2048) == 4096          # 4 x 2048 clamped
    assert M(576).effective_chunk(8192) == 8192          # the caller's own choice stands
    assert M(288).effective_chunk(1024) == 1024 and M(384).effective_chunk(4096) == 4096
    assert M(576).effective_chunk(0) == 4                # (degenerate: at least one cell, scaled)


def test_probe_rule_constants_are_consistent():
    """the load-time probe's acceptance bar in logit units: delta <= RECHECK_MARGIN / PROBE_DIVISOR / (PROBE_FACTOR / 4) = 5.33e-4, and the
    bench's classifiers (delta 1.7e-4 ... 4.4e-4, profiles/r6/probe_vs_real.txt) sit below it"""
    from multiplexed_image_annotator_amd import ops
    v = ops.VitModel
    bar = v.RECHECK_MARGIN / v.PROBE_DIVISOR / (v.PROBE_FACTOR * 0.25)
    assert abs(bar - 5.333e-4) < 1e-6
    assert v.PROBE_FACTOR >= 2.76          # the largest ratio measured (real |dp| over delta / 4)
    assert 4.41e-4 < bar
