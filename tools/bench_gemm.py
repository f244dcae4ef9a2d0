#!/usr/bin/env python3
"""GEMM micro-benchmark on the shapes of the five classifiers (A/B of kernel variants in ONE process, interleaved rounds).
usage: python tools/bench_gemm.py [cells] [variants...]"""
import os

os.environ.setdefault("RIBCA_DIAG", "1")      # the variant / ablation / stamp kernel forms live in libribca_hip_diag.so (build --diag)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from multiplexed_image_annotator_amd import _lib
from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variants = [int(v) for v in sys.argv[2:]] or [1, 2]
# RIBCA_BENCH_ROTATE=n: cycle through n copies of the activation / output buffers so that successive launches do not find
# their operands in the 256 MB Infinity Cache (what a launch sees inside the real pipeline)
ROT = int(os.environ.get("RIBCA_BENCH_ROTATE", "1"))
dev = _lib.require_gpu()
M = cells * 101
shapes = []
for d in (576, 384, 288, 144):
    dp = (d + 31) // 32 * 32
    shapes += [("qkv", d, 3 * d, dp, 0), ("proj", d, d, dp, 0), ("fc1", d, 4 * d, dp, 1), ("fc2", d, d, 4 * d, 0)]
g = torch.Generator(device="cpu").manual_seed(0)
print(f"M = {M} rows ({cells} cells); times in ms, TF = algorithmic TFLOP/s (x3 MFMA passes issued)")
for name, d, n, kp, kind in shapes:
    a = (torch.randn((M, 2 * kp), generator=g) * 0.1).to(torch.bfloat16).view(torch.int16).to(dev)
    a_set = [a] + [a.clone() for _ in range(ROT - 1)]
    npad = lib().ribca_gemm_padded_n(n)
    w = (torch.randn((npad, 2 * kp), generator=g) * 0.1).to(torch.bfloat16).view(torch.int16).to(dev)
    bias = torch.zeros(n, device=dev)
    out = torch.zeros((M, n if kind == 0 else 2 * n), dtype=torch.float32 if kind == 0 else torch.int16, device=dev)
    ldo = n if kind == 0 else 2 * n
    out_set = [out] + [out.clone() for _ in range(ROT - 1)]
    res = {}
    for rnd in range(3):
        for v in variants:
            lib().ribca_set_gemm_variant(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5 if ROT == 1 else 2 * ROT
            check(lib().ribca_test_gemm(kind, ptr(a), 2 * kp, ptr(w), 2 * kp, M, n, kp, ptr(bias), ptr(out), ldo, stream_ptr()), "gemm")
            e0.record()
            for r in range(reps):
                check(lib().ribca_test_gemm(kind, ptr(a_set[r % ROT]), 2 * kp, ptr(w), 2 * kp, M, n, kp, ptr(bias), ptr(out_set[r % ROT]), ldo,
                                            stream_ptr()), "gemm")
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / reps)
    k_alg = d if name != "fc2" else 4 * d
    flops = 2.0 * M * n * k_alg
    line = f"{name:5s} D={d:4d} N={n:5d} K={k_alg:5d}: "
    for v in variants:
        ms = min(res[v])
        line += f" v{v}: {ms:7.3f} ms {flops / ms / 1e9:7.1f} TF |"
    print(line, flush=True)
lib().ribca_set_gemm_variant(0)
