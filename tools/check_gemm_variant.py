#!/usr/bin/env python3
"""Bitwise comparison of a GEMM kernel variant against the production kernel (variant 0) on the same inputs: residual, GELU and
QKV epilogues, ragged M, every column-tile width.  usage: python tools/check_gemm_variant.py VARIANT [VARIANT...]
(tests/test_gpu_kernels.py::test_gemm_duo_variant_bit_identical runs `compare` on a reduced list)"""
import os

os.environ.setdefault("RIBCA_DIAG", "1")      # the variant / ablation / stamp kernel forms live in libribca_hip_diag.so (build --diag)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from multiplexed_image_annotator_amd import _lib
from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr

SHAPES = [(1, 288, 288), (100, 864, 288), (300, 288, 1152), (257, 144, 144), (130, 432, 144), (77, 576, 2304), (200, 1728, 576),
          (129, 384, 384), (64, 64, 64), (1000, 2304, 576), (11100, 576, 576), (22100, 288, 288), (16500, 1536, 384), (40000, 576, 576),
          (70001, 1152, 288), (103424, 576, 144)]
QKV_CASES = ((144, 3), (288, 130), (384, 70), (576, 131))


def _ps(x, kp, dev, rows_pad=None):
    r, k = x.shape
    out = torch.zeros((rows_pad or r, 2 * kp), dtype=torch.int16, device=dev)
    check(lib().ribca_test_pack_weight(ptr(x.contiguous()), r, k, ptr(out), rows_pad or r, kp, stream_ptr()), "pack")
    return out


def compare(variants, shapes=SHAPES, qkv_cases=QKV_CASES, verbose=True):
    """number of (variant, shape, epilogue) cases whose output differs in ANY bit from the production kernel's"""
    dev = _lib.require_gpu()
    g = torch.Generator(device="cpu").manual_seed(1)
    say = print if verbose else (lambda *a, **k: None)
    bad = 0
    try:
        for v in variants:
            for m, n, k in shapes:
                kp = (k + 31) // 32 * 32
                a = torch.randn((m, k), generator=g).to(dev)
                w = (torch.randn((n, k), generator=g) / np.sqrt(k)).to(dev)
                bias = (torch.randn((n,), generator=g) * 0.1).to(dev)
                z0 = torch.randn((m, n), generator=g).to(dev)
                a_ps, w_ps = _ps(a, kp, dev), _ps(w, kp, dev, lib().ribca_gemm_padded_n(n))
                for kind in (0, 1):
                    outs = []
                    for var in (0, v):
                        lib().ribca_set_gemm_variant(var)
                        if kind == 0:
                            out, ldo = z0.clone(), n
                        else:
                            out, ldo = torch.zeros((m, 2 * n), dtype=torch.int16, device=dev), 2 * n
                        check(lib().ribca_test_gemm(kind, ptr(a_ps), 2 * kp, ptr(w_ps), 2 * kp, m, n, kp, ptr(bias), ptr(out), ldo, stream_ptr()), "gemm")
                        torch.cuda.synchronize()
                        outs.append(out.view(torch.int32) if kind == 0 else out)
                    same = torch.equal(outs[0], outs[1])
                    bad += 0 if same else 1
                    if same:
                        say(f"ok v{v} kind={kind} {m}x{n}x{k}", flush=True)
                    else:
                        nz = torch.nonzero(outs[0] != outs[1])
                        say(f"MISMATCH v{v} kind={kind} {m}x{n}x{k}: {len(nz)} elements, first at {nz[0].tolist()}", flush=True)
            # QKV epilogue through the attention test entry point
            for d, cells in qkv_cases:
                heads, ntok = 12, 101
                hd = d // heads
                hdp, hdv = (hd + 7) // 8 * 8, (hd + 15) // 16 * 16
                m = cells * ntok
                dp = (d + 31) // 32 * 32
                y = torch.randn((m, d), generator=g).to(dev)
                w = (torch.randn((3 * d, d), generator=g) / np.sqrt(d)).to(dev)
                bias = (torch.randn((3 * d,), generator=g) * 0.1).to(dev)
                y_ps, w_ps = _ps(y, dp, dev), _ps(w, dp, dev, lib().ribca_gemm_padded_n(3 * d))
                res = []
                for var in (0, v):
                    lib().ribca_set_gemm_variant(var)
                    q = torch.zeros((cells, heads, 112, 2 * hdp), dtype=torch.int16, device=dev)
                    kk = torch.zeros_like(q)
                    vt = torch.zeros_like(q)
                    out = torch.zeros((m, 2 * dp), dtype=torch.int16, device=dev)
                    check(lib().ribca_test_qkv_attention(ptr(y_ps), 2 * dp, ptr(w_ps), 2 * dp, cells, d, dp, ptr(bias), ptr(q), ptr(kk), ptr(vt), ptr(out),
                                                         2 * dp, stream_ptr()), "qkv")
                    torch.cuda.synchronize()
                    res.append((q, kk, vt, out))
                same = all(torch.equal(x, y2) for x, y2 in zip(res[0], res[1]))
                bad += 0 if same else 1
                say(("ok" if same else "MISMATCH") + f" v{v} qkv d={d} cells={cells}", flush=True)
    finally:
        lib().ribca_set_gemm_variant(0)
    return bad


if __name__ == "__main__":
    n_bad = compare([int(v) for v in sys.argv[1:]] or [40])
    print("RESULT:", "all identical" if n_bad == 0 else f"{n_bad} mismatching cases")
    sys.exit(0 if n_bad == 0 else 1)
