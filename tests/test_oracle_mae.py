"""Oracle MAE imputer restatement vs the reference's own MarkerImputer.impute (golden, full-depth seeded weights)."""
import os

import numpy as np
import pytest
import torch

from multiplexed_image_annotator_amd import synth
from oracle import ref_mae


def mae_inputs(panel, n, seed):
    L = synth.MAE_PANELS[panel]
    u = synth.uniform(synth.stream_key(seed, "maex/" + panel), n * L * 1600).reshape(n, L, 40, 40).to(torch.float32)
    x = u * 2 - 1
    return torch.where(x > 0.0, x, torch.full_like(x, -1.0))


@pytest.mark.parametrize("panel", ["immune_base", "immune_full"])
def test_mae_matches_reference(golden_dir, panel):
    g = np.load(os.path.join(golden_dir, "mae.npz"))
    present = g[panel + "_present"].tolist()
    seed = synth.SEED_BASE + 301
    sd = synth.make_mae_state_dict(panel, seed)
    x = mae_inputs(panel, 6, seed)
    L = x.shape[1]
    missing = [c for c in range(L) if c not in present]
    x[:, missing] = -1.0
    torch.set_num_threads(min(8, len(os.sched_getaffinity(0))))
    y = ref_mae.impute(sd, x, present, batch_size=4)
    assert torch.equal(y[:, present], x[:, present])                  # kept channels pass through untouched
    np.testing.assert_allclose(y[:, missing].numpy(), g[panel + "_pred"], rtol=0, atol=5e-5)
    assert np.abs(g[panel + "_pred"]).max() > 0.5                     # predictions are not degenerate
