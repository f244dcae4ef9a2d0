#!/usr/bin/env python3
"""Where does a GEMM workgroup spend its time?  Runs the production kernel with diagnostic stamps (variant 12) on one
classifier shape and prints, per workgroup: entry -> first stage landed (prologue), K loop, epilogue (incl. store accept),
and the gap between consecutive workgroups on the same CU."""
import os
import sys

os.environ.setdefault("RIBCA_DIAG", "1")      # the stamp / ablation kernel forms live in libribca_hip_diag.so (build --diag)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from multiplexed_image_annotator_amd import _lib
from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr

dev = _lib.require_gpu()
cells = 1024
M = cells * 101
g = torch.Generator().manual_seed(0)
# kind 0: fp32 residual epilogue (imputer), 1: GELU, 2: packed-split residual epilogue with row statistics (the classifiers' proj / fc2)
for name, d, n, kp, kind in (("proj576", 576, 576, 576, 0), ("proj576_ps", 576, 576, 576, 2), ("proj288_ps", 288, 288, 288, 2), ("fc2_576_ps", 576, 576, 2304, 2),
                             ("fc1_576", 576, 2304, 576, 1), ("fc2_576", 576, 576, 2304, 0), ("fc1_288", 288, 1152, 288, 1)):
    a = (torch.randn((M, 2 * kp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev)
    npad = lib().ribca_gemm_padded_n(n)
    w = (torch.randn((npad, 2 * kp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev)
    bias = torch.zeros(n, device=dev)
    out = torch.zeros((M, n if kind == 0 else 2 * n), dtype=torch.float32 if kind == 0 else torch.int16, device=dev)
    ldo = n if kind == 0 else 2 * n
    if kind == 2:
        part = torch.zeros((lib().ribca_test_resid_tiles(n), M, 2), dtype=torch.float32, device=dev)
        rs = torch.zeros((M, 2), dtype=torch.float32, device=dev)
        prev = torch.zeros((M, 2), dtype=torch.float32, device=dev)
    bn = 128 if n % 128 == 0 else (96 if n % 96 == 0 else 64)
    nblk = ((M + 255) // 256) * (npad // bn)
    stamps = torch.zeros((nblk, 20), dtype=torch.int64, device=dev)
    lib().ribca_set_gemm_stamps(ptr(stamps), nblk)
    lib().ribca_set_gemm_variant(12)
    for _ in range(2):
        if kind == 2:
            check(lib().ribca_test_gemm_resid_ps(ptr(a), 2 * kp, ptr(w), 2 * kp, M, n, kp, ptr(bias), ptr(out), ldo, ptr(part), ptr(rs), ptr(prev), stream_ptr()),
                  "resid_ps")
        else:
            check(lib().ribca_test_gemm(kind, ptr(a), 2 * kp, ptr(w), 2 * kp, M, n, kp, ptr(bias), ptr(out), ldo, stream_ptr()), "gemm")
    torch.cuda.synchronize()
    lib().ribca_set_gemm_variant(0)
    lib().ribca_set_gemm_stamps(None, 0)
    t = stamps.cpu().numpy().astype(np.float64)
    t0, t1, t2, t3 = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    us = 0.01  # 100 MHz ticks -> microseconds
    span = (t3.max() - t0.min()) * us
    cu = (t[:, 4].astype(np.int64) << 32) | (t[:, 5].astype(np.int64) & 0xFFFFFF00)   # XCC + (SE, SH, CU) bits of HW_ID
    gaps = []
    for key in np.unique(cu):
        sel = np.flatnonzero(cu == key)
        order = sel[np.argsort(t0[sel])]
        gaps += list((t0[order][1:] - t3[order][:-1]) * us)
    clk = (t[:, 19] - t[:, 18]) / np.maximum((t2 - t1), 1) * 0.1      # shader cycles per 10 ns tick -> GHz
    print(f"{name}: {nblk} workgroups, kernel span {span:.1f} us, nk={kp // 32}, in-kernel clock {np.median(clk):.2f} GHz (K loops)")
    for label, v in (("prologue  (entry -> first stage landed)", (t1 - t0) * us), ("K loop", (t2 - t1) * us),
                     ("epilogue  (loop end -> stores accepted)", (t3 - t2) * us), ("workgroup total", (t3 - t0) * us),
                     ("gap to next workgroup on the same CU", np.array(gaps))):
        if len(v):
            print(f"   {label:44s} median {np.median(v):7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f} us")
    ends = (t[:, 6:18] - t2[:, None]) * us            # every wave's "stores accepted" relative to the end of wave 0's K loop
    print("   per-wave epilogue end after K loop (median us): consumers " + " ".join(f"{np.median(ends[:, w]):.2f}" for w in range(8))
          + " | loaders " + " ".join(f"{np.median(ends[:, w]):.2f}" for w in range(8, 12)))
    last = ends.max(axis=1)
    gaps2 = []
    for key in np.unique(cu):
        sel = np.flatnonzero(cu == key)
        order = sel[np.argsort(t0[sel])]
        gaps2 += list((t0[order][1:] - (t2[order][:-1] + last[order][:-1] / us)) * us)
    print(f"   last wave done {np.median(last):.2f} us after the K loop; next workgroup enters {np.median(gaps2):.2f} us after that (median)")
