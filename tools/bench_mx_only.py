#!/usr/bin/env python3
"""times the MX residual GEMM alone (packed operands, statistics included) for the fc2 shapes; RIBCA_LIB selects a timing-ablation build
(tools/build_mx_variant.py).  usage: python tools/bench_mx_only.py [cells] [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multiplexed_image_annotator_amd import _lib
from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tag = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("RIBCA_LIB", "product")
ROT = 4
dev = _lib.require_gpu()
M = cells * 101
g = torch.Generator(device="cpu").manual_seed(0)
out = []
for d, k in ((576, 2304), (384, 1536), (288, 1152), (384, 384)):
    dp = (d + 31) // 32 * 32
    w = (torch.randn((lib().ribca_gemm_padded_n(d), 2 * k), generator=g) * 0.05).to(torch.float16).view(torch.int16).to(dev)
    bias = torch.zeros(d, device=dev)
    z_set = [(torch.randn((M, 2 * dp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev) for _ in range(ROT)]
    part = torch.zeros((d // 48, M, 2), device=dev); rs = torch.zeros((M, 2), device=dev); prev = torch.zeros((M, 2), device=dev)
    mx_set = []
    wh = torch.zeros(lib().ribca_test_mx_weight_bytes(d, k, 0), dtype=torch.uint8, device=dev)
    wx = torch.zeros(lib().ribca_test_mx_weight_bytes(d, k, 1), dtype=torch.uint8, device=dev)
    for r in range(ROT):
        a = (torch.randn((M, 2 * k), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev)
        hi = torch.zeros((M, k), dtype=torch.int16, device=dev); l8 = torch.zeros((M, k), dtype=torch.uint8, device=dev)
        sc = torch.zeros((M, k // 32), dtype=torch.uint8, device=dev)
        if r == 0:
            check(lib().ribca_test_gemm_mx_resid(ptr(a), 2 * k, ptr(w), 2 * k, 128, d, k, ptr(bias), ptr(hi), ptr(l8), ptr(sc), ptr(wh), ptr(wx),
                                                 ptr(z_set[0]), 2 * dp, None, None, None, stream_ptr()), "w")
        check(lib().ribca_test_mx_pack_act(ptr(a), 2 * k, M, k, ptr(hi), ptr(l8), ptr(sc), stream_ptr()), "pack")
        mx_set.append((hi, l8, sc)); del a
    best = 1e9
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 * ROT
        torch.cuda.synchronize(); e0.record()
        for r in range(reps):
            hi, l8, sc = mx_set[r % ROT]
            check(lib().ribca_test_gemm_mx_resid_packed(ptr(hi), ptr(l8), ptr(sc), k, ptr(wh), ptr(wx), M, d, ptr(bias), ptr(z_set[r % ROT]), 2 * dp,
                                                        ptr(part), ptr(rs), ptr(prev), stream_ptr()), "mx")
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    out.append(f"D={d} K={k}: {best:.3f} ms ({2.0 * M * d * k / best / 1e9:.0f} TF)")
    del z_set, mx_set
print(f"{tag:24s} " + " | ".join(out), flush=True)
