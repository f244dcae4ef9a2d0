"""The diagnostic library (-DRIBCA_DIAG: the A/B, ablation and stamp kernel forms tools/ drive) must keep compiling when the product
kernels change: a syntax-only pass instantiates every template of the translation units that carry such forms (seconds, no code
generation; the full build is `python -m multiplexed_image_annotator_amd.build --diag`)."""
import os
import subprocess

import pytest

from multiplexed_image_annotator_amd import build as B


@pytest.mark.parametrize("src", ["gemm_duo.hip", "gemm_split16.hip", "cell_attention.hip", "ribca_api.hip"])
def test_diag_translation_unit_compiles(src):
    cmd = [B._hipcc()] + B.FLAGS + ["-DRIBCA_DIAG", "-fsyntax-only", os.path.join(B.CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
