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
