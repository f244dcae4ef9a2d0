#!/usr/bin/env python3
"""Is the GEMM epilogue limited per CU (a latency chain) or chip-wide (HBM write bandwidth shared by every CU that is in its
epilogue at the same moment)?  One wave of tiles (<= 256 workgroups, one per CU) of the fc1 / proj shapes at D = 576 with a growing
number of active CUs; the launch time is then one tile's time.  Variant 9 = same kernel without the epilogue."""
import os

os.environ.setdefault("RIBCA_DIAG", "1")      # the variant / ablation / stamp kernel forms live in libribca_hip_diag.so (build --diag)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from multiplexed_image_annotator_amd import _lib
from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr

dev = _lib.require_gpu()
g = torch.Generator(device="cpu").manual_seed(0)
for name, n, k, kind in (("fc1", 2304, 576, 1), ("proj", 576, 576, 0), ("fc1_288", 1152, 288, 1)):
    kp = k
    npad = lib().ribca_gemm_padded_n(n)
    w = (torch.randn((npad, 2 * kp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev)
    bias = torch.zeros(n, device=dev)
    ntiles = npad // (128 if n % 128 == 0 else 96)
    for mt in (1, 2, 4, 7, 10, 14, 28, 56):
        M = 256 * mt
        a = (torch.randn((M, 2 * kp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev)
        out = torch.zeros((M, n if kind == 0 else 2 * n), dtype=torch.float32 if kind == 0 else torch.int16, device=dev)
        ldo = n if kind == 0 else 2 * n
        line = f"{name} N={n} K={k} m-tiles {mt:3d} -> {mt * ntiles:5d} workgroups: "
        for v in (0, 9):
            lib().ribca_set_gemm_variant(v)
            best = 1e9
            for rep in range(5):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    check(lib().ribca_test_gemm(kind, ptr(a), 2 * kp, ptr(w), 2 * kp, M, n, kp, ptr(bias), ptr(out), ldo, stream_ptr()), "gemm")
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 4)
            line += f" v{v}: {best * 1e3:7.1f} us |"
        print(line, flush=True)
lib().ribca_set_gemm_variant(0)
