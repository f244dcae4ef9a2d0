#!/usr/bin/env python3
"""mlp.fc2 (and attn.proj) of the classifiers: the fp16x3 kernels (one workgroup per CU with the LDS drain; two per CU with the residual
tile through the ring) against the MX kernel (csrc/gemm_mx.hip: fp16 hi*hi + two block-scaled corrections, 3-byte activations), interleaved
rounds in ONE process, row statistics (ln_finalize) included on every side, operands rotated through ROT buffer sets (cache-cold).
The MX operands are packed outside the timed region (the forward's producers emit them); the duo hook re-packs its weight on every call.
usage: python tools/bench_mx.py [cells]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from multiplexed_image_annotator_amd import _lib
from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ROT = int(os.environ.get("RIBCA_BENCH_ROTATE", "4"))
dev = _lib.require_gpu()
M = cells * 101
g = torch.Generator(device="cpu").manual_seed(0)
print(f"M = {M} rows ({cells} cells), {ROT} buffer sets; ms per launch (best of 3 rounds), TF = algorithmic TFLOP/s")
tot = {"split": 0.0, "duo": 0.0, "mx": 0.0}
for d in (576, 384, 288):
    dp = (d + 31) // 32 * 32
    for name, k in (("fc2", 4 * d), ("proj", d)):
        if k % 128 != 0:
            continue
        kp = k
        a_set = [(torch.randn((M, 2 * kp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev) for _ in range(ROT)]
        npad = lib().ribca_gemm_padded_n(d)
        w = (torch.randn((npad, 2 * kp), generator=g) * 0.05).to(torch.float16).view(torch.int16).to(dev)
        wf = torch.zeros_like(w)
        bias = torch.zeros(d, device=dev)
        z_set = [(torch.randn((M, 2 * dp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev) for _ in range(ROT)]
        part = torch.zeros((max(lib().ribca_test_resid_part_rows(d), d // 48), M, 2), device=dev)
        rs = torch.zeros((M, 2), device=dev)
        prev = torch.zeros((M, 2), device=dev)
        # MX operands
        mx_set = []
        for r in range(ROT):
            hi = torch.zeros((M, kp), dtype=torch.int16, device=dev); l8 = torch.zeros((M, kp), dtype=torch.uint8, device=dev)
            sc = torch.zeros((M, kp // 32), dtype=torch.uint8, device=dev)
            check(lib().ribca_test_mx_pack_act(ptr(a_set[r]), 2 * kp, M, kp, ptr(hi), ptr(l8), ptr(sc), stream_ptr()), "pack")
            mx_set.append((hi, l8, sc))
        wh = torch.zeros(lib().ribca_test_mx_weight_bytes(d, kp, 0), dtype=torch.uint8, device=dev)
        wx = torch.zeros(lib().ribca_test_mx_weight_bytes(d, kp, 1), dtype=torch.uint8, device=dev)
        # (the weight image through the test hook of the unpacked entry point, once)
        hi0, l80, sc0 = mx_set[0]
        check(lib().ribca_test_gemm_mx_resid(ptr(a_set[0]), 2 * kp, ptr(w), 2 * kp, 128, d, kp, ptr(bias), ptr(hi0), ptr(l80), ptr(sc0), ptr(wh), ptr(wx),
                                             ptr(z_set[0]), 2 * dp, None, None, None, stream_ptr()), "mx weight")
        res = {"split": [], "duo": [], "mx": []}

        def run(which, r):
            if which == "split":
                check(lib().ribca_test_gemm_resid_ps(ptr(a_set[r]), 2 * kp, ptr(w), 2 * kp, M, d, kp, ptr(bias), ptr(z_set[r]), 2 * dp, ptr(part),
                                                     ptr(rs), ptr(prev), stream_ptr()), "split")
            elif which == "duo":
                check(lib().ribca_test_gemm_resid_ps_duo(ptr(a_set[r]), 2 * kp, ptr(w), 2 * kp, M, d, kp, ptr(bias), ptr(wf), ptr(z_set[r]), 2 * dp,
                                                         ptr(part), ptr(rs), ptr(prev), stream_ptr()), "duo")
            else:
                hi, l8, sc = mx_set[r]
                check(lib().ribca_test_gemm_mx_resid_packed(ptr(hi), ptr(l8), ptr(sc), kp, ptr(wh), ptr(wx), M, d, ptr(bias), ptr(z_set[r]), 2 * dp,
                                                            ptr(part), ptr(rs), ptr(prev), stream_ptr()), "mx")

        for rnd in range(3):
            for which in ("split", "duo", "mx"):
                run(which, 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 2 * ROT
                e0.record()
                for r in range(reps):
                    run(which, r % ROT)
                e1.record()
                torch.cuda.synchronize()
                res[which].append(e0.elapsed_time(e1) / reps)
        flops = 2.0 * M * d * k
        ms = {kk: min(v) for kk, v in res.items()}
        for kk in tot:
            tot[kk] += ms[kk]
        best_old = min(ms["split"], ms["duo"])
        print(f"{name:5s} D={d:4d} K={k:5d}: split {ms['split']:7.3f} ms {flops / ms['split'] / 1e9:6.1f} TF | duo+ring {ms['duo']:7.3f} ms "
              f"{flops / ms['duo'] / 1e9:6.1f} TF | MX {ms['mx']:7.3f} ms {flops / ms['mx'] / 1e9:6.1f} TF | MX vs best fp16x3 {100.0 * (ms['mx'] / best_old - 1.0):+6.1f} %",
              flush=True)
        del a_set, z_set, mx_set
print(f"sum: split {tot['split']:.3f} ms, duo+ring {tot['duo']:.3f} ms, MX {tot['mx']:.3f} ms")
