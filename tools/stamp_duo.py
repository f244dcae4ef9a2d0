#!/usr/bin/env python3
"""Per-CU timelines of the two-workgroups-per-CU GEMM (variant 44 = gemm_duo.hip with stamps): phases per workgroup and how much of
a workgroup's epilogue / prologue runs beside a co-resident workgroup's K loop."""
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
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 48
for name, d, n, kp, kind in (("proj576", 576, 576, 576, 0), ("fc1_576", 576, 2304, 576, 1), ("fc2_576", 576, 576, 2304, 0), ("fc1_288", 288, 1152, 288, 1)):
    a = (torch.randn((M, 2 * kp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev)
    npad = lib().ribca_gemm_padded_n(n)
    w = (torch.randn((npad, 2 * kp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev)
    bias = torch.zeros(n, device=dev)
    out = torch.zeros((M, n if kind == 0 else 2 * n), dtype=torch.float32 if kind == 0 else torch.int16, device=dev)
    ldo = n if kind == 0 else 2 * n
    bn = 128 if n % 128 == 0 else (96 if n % 96 == 0 else 64)
    # the default duo form (RIBCA_DUO_FORM 0 / 1) launches 192-row tiles, form 2 256-row tiles; the library also refuses to stamp
    # workgroups beyond the capacity it is told
    bm = 256 if os.environ.get("RIBCA_DUO_FORM") == "2" or not (40 <= variant <= 49) else 192
    nblk = ((M + bm - 1) // bm) * (npad // bn)
    stamps = torch.zeros((nblk, 20), dtype=torch.int64, device=dev)
    lib().ribca_set_gemm_stamps(ptr(stamps), nblk)
    lib().ribca_set_gemm_variant(variant)
    for _ in range(2):
        check(lib().ribca_test_gemm(kind, ptr(a), 2 * kp, ptr(w), 2 * kp, M, n, kp, ptr(bias), ptr(out), ldo, stream_ptr()), "gemm")
    torch.cuda.synchronize()
    lib().ribca_set_gemm_variant(0)
    lib().ribca_set_gemm_stamps(None, 0)
    t = stamps.cpu().numpy().astype(np.float64)
    us = 0.01
    t0, t1, t2 = t[:, 0] * us, t[:, 1] * us, t[:, 2] * us
    t3 = t[:, 6:10].max(axis=1) * us
    base = t0.min()
    span = t3.max() - base
    cu = (t[:, 4].astype(np.int64) << 32) | (t[:, 5].astype(np.int64) & 0xFF00)      # XCC + (SE, SH, CU) bits of HW_ID
    clk = (t[:, 19] - t[:, 18]) / np.maximum((t[:, 2] - t[:, 1]), 1) * 0.1      # shader cycles per 10 ns tick -> GHz
    print(f"{name}: {nblk} workgroups on {len(np.unique(cu))} CUs, kernel span {span:.1f} us, nk={kp // 32}, in-kernel clock {np.median(clk):.2f} GHz (K loops)")
    for label, v in (("prologue (entry -> first stage landed)", t1 - t0), ("K loop", t2 - t1), ("epilogue (loop end -> last wave's stores accepted)", t3 - t2),
                     ("workgroup total", t3 - t0)):
        print(f"   {label:52s} median {np.median(v):7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f} us")
    # per CU: time with 0 / 1 / 2 workgroups in their K loop, and resident workgroups
    kl = np.zeros(3)
    res = np.zeros(4)
    for key in np.unique(cu):
        sel = np.flatnonzero(cu == key)
        ev = []
        for i in sel:
            ev += [(t1[i], 0, +1), (t2[i], 0, -1), (t0[i], 1, +1), (t3[i], 1, -1)]
        ev.sort()
        nk_, nr, last = 0, 0, base
        for tt, kind_, dlt in ev:
            kl[min(nk_, 2)] += tt - last
            res[min(nr, 3)] += tt - last
            last = tt
            if kind_ == 0: nk_ += dlt
            else: nr += dlt
        kl[0] += (base + span) - last
        res[0] += (base + span) - last
    kl /= kl.sum(); res /= res.sum()
    print(f"   CU time with 0 / 1 / 2 workgroups in their K loop: {kl[0]:.2f} / {kl[1]:.2f} / {kl[2]:.2f};  resident 0 / 1 / 2 / 3+: " + " / ".join(f"{x:.2f}" for x in res))
    # one CU's first workgroups as a timeline
    key = np.unique(cu)[0]
    sel = np.flatnonzero(cu == key)
    sel = sel[np.argsort(t0[sel])][:12]
    for i in sel:
        print(f"      wg {i:6d}: entry {t0[i] - base:7.2f}  first stage {t1[i] - base:7.2f}  loop end {t2[i] - base:7.2f}  done {t3[i] - base:7.2f}   tg slot {(int(t[i,5]) >> 16) & 15}")
