#!/usr/bin/env python3
"""Vendor yardstick, OUTSIDE the product path (never imported by the package; bench.py's roofline block and this script are its only
callers): what torch.matmul (hipBLASLt / rocBLAS) sustains in plain fp16 / bf16 on the dominant GEMM shapes of the classifiers (reference
op: timm Mlp.fc1 / fc2, Attention.qkv / proj reached from cell_type_annotation/model.py:54-55), on the same box, with operands rotated so
that nothing is served from a warm cache.  One pass of a plain 16-bit GEMM is ONE matrix unit; the product kernels issue 1.75 (MX) or 3
(fp16x3) units per product, so the figure to compare with is a product kernel's ISSUED rate (algorithmic TFLOP/s x units), not its
algorithmic one.
  python tools/vendor_gemm.py [cells]      ->  one JSON line (TFLOP/s, ms and operand + result GB/s per shape and dtype)"""
import json
import sys

SHAPES = [("fc2@576", 2304, 576), ("fc1@576", 576, 2304), ("qkv@576", 576, 1728), ("fc2@384", 1536, 384), ("fc1@384", 384, 1536),
          ("fc2@288", 1152, 288), ("fc1@288", 288, 1152), ("proj@576", 576, 576), ("proj@288", 288, 288)]


def measure(cells: int, names=None, dtypes=("fp16",), rot: int = 3, rounds: int = 3):
    """{dtype: {shape name: {"M", "K", "N", "ms", "tflops", "hbm_tb_s"}}} for C[M, N] = A[M, K] W[N, K]^T with M = cells * 101"""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    M = cells * 101
    out = {}
    for dt_name in dtypes:
        dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[dt_name]
        res = {}
        for name, K, N in SHAPES:
            if names is not None and name not in names:
                continue
            a = [(torch.randn((M, K), device=dev) * 0.1).to(dt) for _ in range(rot)]
            w = (torch.randn((N, K), device=dev) * 0.05).to(dt)
            c = [torch.empty((M, N), device=dev, dtype=dt) for _ in range(rot)]
            for r in range(rot):
                torch.matmul(a[r], w.t(), out=c[r])
            best = 1e9
            for _ in range(rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 4 * rot
                torch.cuda.synchronize()
                e0.record()
                for r in range(reps):
                    torch.matmul(a[r % rot], w.t(), out=c[r % rot])
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps)
            res[name] = {"M": M, "K": K, "N": N, "ms": round(best, 4), "tflops": round(2.0 * M * N * K / best / 1e9, 1),
                         "hbm_tb_s": round((M * K + M * N) * 2 / best / 1e9, 2)}
            del a, c
        out[dt_name] = res
    return out


if __name__ == "__main__":
    import torch
    cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    print(json.dumps({"vendor_gemm": measure(cells, dtypes=("fp16", "bf16")), "cells": cells, "torch": torch.__version__}))
