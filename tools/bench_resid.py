#!/usr/bin/env python3
"""Residual GEMMs (attn.proj, mlp.fc2) of the five classifiers on the packed-split stream: the one-workgroup-per-CU kernel with the LDS
drain (EpiResidPS) against the two-workgroups-per-CU kernel with the residual tile riding the A ring (EpiResidZK), interleaved rounds in
ONE process, row statistics (ln_finalize) included on both sides.  The duo hook re-packs the fragment-order weight on every call (a
5-20 us launch the forward does not pay).
usage: python tools/bench_resid.py [cells]        RIBCA_BENCH_ROTATE=n: n copies of the operand / residual buffers (cache-cold, default 4)"""
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
tot = {"split": 0.0, "duo": 0.0}
for d in (576, 384, 288, 144):
    dp = (d + 31) // 32 * 32
    for name, k in (("proj", d), ("fc2", 4 * d)):
        kp = (k + 31) // 32 * 32
        a_set = [(torch.randn((M, 2 * kp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev) for _ in range(ROT)]
        npad = lib().ribca_gemm_padded_n(d)
        w = (torch.randn((npad, 2 * kp), generator=g) * 0.05).to(torch.float16).view(torch.int16).to(dev)
        wf = torch.zeros_like(w)
        bias = torch.zeros(d, device=dev)
        z_set = [(torch.randn((M, 2 * dp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev) for _ in range(ROT)]
        part = torch.zeros((lib().ribca_test_resid_part_rows(d), M, 2), device=dev)
        rs = torch.zeros((M, 2), device=dev)
        prev = torch.zeros((M, 2), device=dev)
        res = {"split": [], "duo": []}

        def run(which, r):
            if which == "split":
                check(lib().ribca_test_gemm_resid_ps(ptr(a_set[r]), 2 * kp, ptr(w), 2 * kp, M, d, kp, ptr(bias), ptr(z_set[r]), 2 * dp, ptr(part),
                                                     ptr(rs), ptr(prev), stream_ptr()), "split")
            else:
                check(lib().ribca_test_gemm_resid_ps_duo(ptr(a_set[r]), 2 * kp, ptr(w), 2 * kp, M, d, kp, ptr(bias), ptr(wf), ptr(z_set[r]), 2 * dp,
                                                         ptr(part), ptr(rs), ptr(prev), stream_ptr()), "duo")

        for rnd in range(3):
            for which in ("split", "duo"):
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
        ms_s, ms_d = min(res["split"]), min(res["duo"])
        tot["split"] += ms_s
        tot["duo"] += ms_d
        print(f"{name:5s} D={d:4d} K={k:5d}: split {ms_s:7.3f} ms {flops / ms_s / 1e9:6.1f} TF | duo+ring {ms_d:7.3f} ms {flops / ms_d / 1e9:6.1f} TF | "
              f"{100.0 * (ms_d / ms_s - 1.0):+6.1f} %", flush=True)
        del a_set, z_set
print(f"sum over the eight shapes: split {tot['split']:.3f} ms, duo+ring {tot['duo']:.3f} ms ({100.0 * (tot['duo'] / tot['split'] - 1.0):+.1f} %)")
