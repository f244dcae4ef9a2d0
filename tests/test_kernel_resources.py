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


@pytest.mark.parametrize("src", ["gemm_duo.hip", "gemm_mx.hip", "gemm_split16.hip", "cell_attention.hip"])
def test_no_lds_read_in_flight_across_a_ring_barrier(src, tmp_path):
    """The K loops of these kernels refill an LDS ring slot right behind the barrier that declares it read.  That holds only if every
    ds_read of the slot has RETURNED (s_waitcnt lgkmcnt(0)) before the wave reaches the barrier: a read merely issued in front of it is
    protected by nothing but latencies (round 4: hipcc had pipelined the fused cell kernel's loop that way -- its fragment reads were
    waited for BEHIND the barrier -- which is what made a faster loader non-repeatable in round 3).  Checked in the ISA: walking back from
    every s_barrier, an s_waitcnt with lgkmcnt(0) must come before any ds_read."""
    asm = tmp_path / "k.s"
    cmd = [B._hipcc()] + [f for f in B.FLAGS if f != "-fPIC"] + ["--cuda-device-only", "-S", os.path.join(B.CSRC, src), "-o", str(asm)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    bad, barriers = [], 0
    kernel = None
    lines = [l.strip() for l in open(asm) if l.strip() and not l.strip().startswith(";")]
    for i, l in enumerate(lines):
        m = re.match(r"(_ZN5ribca\w+):", l)
        if m:
            kernel = m.group(1)
        if not l.startswith("s_barrier"):
            continue
        barriers += 1
        for j in range(i - 1, max(i - 400, -1), -1):
            if lines[j].startswith("s_waitcnt") and "lgkmcnt(0)" in lines[j]:
                break
            if lines[j].startswith("s_barrier") or lines[j].endswith(":") and lines[j].startswith("_ZN"):
                break
            if lines[j].startswith("ds_read"):
                bad.append((kernel, i, lines[j]))
                break
    assert barriers > 0
    assert not bad, bad[:5]
