"""Kernels that fill registers with inline-asm loads (global_load / ds_read / LDS-DMA behind counted s_waitcnt) must compile without
register spills and without scratch: the compiler does not know those registers are still being filled, so a spill stores garbage
(round 3: a 192 x 192 tile form of gemm_duo.hip gave wrong bits that way), and any scratch access is a vector-memory operation the
counted vmcnt waits know nothing of (round 3: hipcc merged two LDS-DMA paths into one load whose buffer descriptor it re-read from
scratch every K step).  hipcc reports both per kernel with -Rpass-analysis=kernel-resource-usage; no GPU needed."""
import os
import re
import subprocess

import pytest

from multiplexed_image_annotator_amd import build as B


@pytest.mark.parametrize("src", ["gemm_duo.hip", "gemm_mx.hip", "gemm_split16.hip", "cell_attention.hip", "attention.hip"])
def test_no_spills_no_scratch(src, tmp_path):
    cmd = [B._hipcc()] + B.FLAGS + ["--cuda-device-only", "-c", os.path.join(B.CSRC, src), "-o", str(tmp_path / "x.o"),
                                    "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    name, seen, bad = None, 0, []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            seen += 1
            continue
        m = re.search(r"(VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and int(m.group(2)) != 0:
            bad.append((name, m.group(1), int(m.group(2))))
    assert seen > 0, "no kernel-resource-usage remarks in the compiler output"
    assert not bad, bad
