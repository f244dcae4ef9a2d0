#!/usr/bin/env python3
"""where a wave of the MX kernel spends its K loop: shader-clock cycles per phase (timing build: python tools/build_mx_variant.py
libribca_mx_stamp.so -DMXDBG_STAMP ; RIBCA_LIB=libribca_mx_stamp.so python tools/stamp_mx.py [cells]).  The variant writes the sums behind
the statistics partials; shapes: fc2 of the MX widths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multiplexed_image_annotator_amd import _lib
from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = _lib.require_gpu()
M = cells * 101
g = torch.Generator(device="cpu").manual_seed(0)
for d, k in ((576, 2304), (384, 1536), (576, 640)):
    dp = (d + 31) // 32 * 32
    w = (torch.randn((lib().ribca_gemm_padded_n(d), 2 * k), generator=g) * 0.05).to(torch.float16).view(torch.int16).to(dev)
    bias = torch.zeros(d, device=dev)
    z = (torch.randn((M, 2 * dp), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev)
    nblk = ((M + 127) // 128) * ((d + 191) // 192)
    part = torch.zeros(((d // 48) * M + nblk * 16, 2), device=dev)
    rs = torch.zeros((M, 2), device=dev); prev = torch.zeros((M, 2), device=dev)
    wh = torch.zeros(lib().ribca_test_mx_weight_bytes(d, k, 0), dtype=torch.uint8, device=dev)
    wx = torch.zeros(lib().ribca_test_mx_weight_bytes(d, k, 1), dtype=torch.uint8, device=dev)
    a = (torch.randn((M, 2 * k), generator=g) * 0.1).to(torch.float16).view(torch.int16).to(dev)
    hi = torch.zeros((M, k), dtype=torch.int16, device=dev); l8 = torch.zeros((M, k), dtype=torch.uint8, device=dev)
    sc = torch.zeros((M, k // 32), dtype=torch.uint8, device=dev)
    check(lib().ribca_test_gemm_mx_resid(ptr(a), 2 * k, ptr(w), 2 * k, 128, d, k, ptr(bias), ptr(hi), ptr(l8), ptr(sc), ptr(wh), ptr(wx),
                                         ptr(z), 2 * dp, None, None, None, stream_ptr()), "w")
    check(lib().ribca_test_mx_pack_act(ptr(a), 2 * k, M, k, ptr(hi), ptr(l8), ptr(sc), stream_ptr()), "pack")
    warm = int(os.environ.get("RIBCA_STAMP_WARM", "400"))      # back-to-back launches before the one read: the chip's clock settles under load
    for rep in range(warm + 1):      # the last launch is the one read
        part.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().ribca_test_gemm_mx_resid_packed(ptr(hi), ptr(l8), ptr(sc), k, ptr(wh), ptr(wx), M, d, ptr(bias), ptr(z), 2 * dp,
                                                    ptr(part), ptr(rs), ptr(prev), stream_ptr()), "mx")
        e1.record(); torch.cuda.synchronize()
    st = part[(d // 48) * M:].reshape(nblk * 4, 4, 2).cpu().double()
    st = st[st[:, 2, 1] > 0]                      # waves that wrote (live column blocks)
    nb = st[0, 2, 1].item()
    b1, f16, cv, b2, mx, tail = st[:, 0, 0], st[:, 0, 1], st[:, 1, 0], st[:, 1, 1], st[:, 2, 0], st[:, 3, 0]
    mhz = st[:, 3, 1]
    tot = b1 + f16 + cv + b2 + mx
    f = lambda x: f"{(x / nb).mean().item():7.0f}"
    print(f"D={d} K={k}: {e0.elapsed_time(e1):.3f} ms, {len(st)} waves, {nb:.0f} steps; cycles per step (mean over waves): "
          f"B1 wait+barrier {f(b1)} | f16 phase {f(f16)} | conversion {f(cv)} | B2 wait+barrier {f(b2)} | MX phase {f(mx)} | step {f(tot)} ; "
          f"behind the K loop {tail.mean().item():.0f} cycles; in-kernel clock (d s_memtime / d s_memrealtime) median {mhz.median().item():.0f} MHz "
          f"(5 % .. 95 %: {mhz.quantile(0.05).item():.0f} .. {mhz.quantile(0.95).item():.0f})", flush=True)
