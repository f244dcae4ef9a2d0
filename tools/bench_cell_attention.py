#!/usr/bin/env python3
"""Times the per-cell fused qkv + attention kernel (cell_attention.hip) on 1024 cells against the unfused pair, per D.
RIBCA_CELL_DBG=1: no attention phase, 2: no MFMAs / fragment reads in the qkv phase (weight stream + barriers only), 3: both
(results wrong by construction: timing ablations)."""
import os
import sys

os.environ.setdefault("RIBCA_DIAG", "1")      # the RIBCA_CELL_DBG ablations exist in libribca_hip_diag.so only (build --diag)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from multiplexed_image_annotator_amd import _lib
from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr

dev = _lib.require_gpu()
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator().manual_seed(0)
for d in (384, 288, 144):
    dp = (d + 31) // 32 * 32
    m = cells * 101
    hd = d // 12
    hdp = (hd + 7) // 8 * 8
    z = (torch.randn((m, 2 * dp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev)
    npad = lib().ribca_gemm_padded_n(3 * d)
    w = (torch.randn((npad, 2 * dp), generator=g) * 0.05).to(torch.float16).view(torch.int16).to(dev)
    bias = torch.zeros(3 * d, device=dev)
    csum = torch.zeros(3 * d, device=dev)
    rs = torch.ones((m, 2), device=dev)
    out = torch.zeros((m, 2 * dp), dtype=torch.int16, device=dev)
    q = torch.zeros((cells, 12, 112, 2 * hdp), dtype=torch.int16, device=dev)
    k, vt = torch.zeros_like(q), torch.zeros_like(q)

    def fused():
        check(lib().ribca_test_cell_attention(ptr(z), 2 * dp, ptr(w), 2 * dp, cells, d, ptr(bias), ptr(csum), ptr(rs), ptr(out), 2 * dp, stream_ptr()), "fused")

    def unfused():
        check(lib().ribca_test_qkv_attention_fold(ptr(z), 2 * dp, ptr(w), 2 * dp, cells, d, dp, ptr(bias), ptr(csum), ptr(rs), ptr(q), ptr(k), ptr(vt), ptr(out),
                                                  2 * dp, stream_ptr()), "unfused")

    res = {}
    for name, fn in (("fused", fused), ("unfused qkv + attention", unfused)):
        fn()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        res[name] = best
    flops = cells * (2.0 * 101 * d * 3 * d + 4.0 * 101 * 101 * d)
    print(f"D={d} cells={cells} dbg={os.environ.get('RIBCA_CELL_DBG', '0')}: " + "  ".join(f"{k_}: {v * 1e3:.1f} us ({flops / v / 1e9:.0f} TF alg)" for k_, v in res.items()), flush=True)
