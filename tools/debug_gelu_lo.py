#!/usr/bin/env python3
"""(GPU box) where the lo codes of the fused GELU -> MX3 producer differ from the packer's on the packed-split kernel's output: by magnitude of x"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import mx_emulation as mx
import test_gpu_mx as T
from test_gpu_kernels import _fold, _ln_case, _row_stats, rnd, ps_decode
from multiplexed_image_annotator_amd import _lib
from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
dev = _lib.require_gpu()
m, d, mean, std = 7001, 288, 0.5, 3.0
n = 4 * d
z_ps, zq, g, b, dp = _ln_case(m, d, 50, dev, mean, std, row_scale=True)
w = rnd((n, d), 53, dev, 2.0 / np.sqrt(d)); bias = rnd((n,), 54, dev, 0.1)
w_ps, csum, bias2 = _fold(w, g, b, bias, dp, dev)
rs = _row_stats(z_ps, dp, m, d, dev)
wf = torch.zeros_like(w_ps)
hi_p, l8_p, sc_p = T._planes(m, n, dev)
check(lib().ribca_test_gemm_gelu_mx(ptr(z_ps), 2 * dp, ptr(w_ps), 2 * dp, m, n, dp, ptr(bias2), ptr(csum), ptr(rs), ptr(wf), ptr(hi_p), ptr(l8_p), ptr(sc_p), stream_ptr()), "gelu_mx")
out = torch.zeros((m, 2 * n), dtype=torch.int16, device=dev)
check(lib().ribca_test_gemm_fold(1, ptr(z_ps), 2 * dp, ptr(w_ps), 2 * dp, m, n, dp, ptr(bias2), ptr(csum), ptr(rs), ptr(out), 2 * n, stream_ptr()), "gemm_fold")
hi_r, l8_r, sc_r = T._planes(m, n, dev)
check(lib().ribca_test_mx_pack_act(ptr(out), 2 * n, m, n, ptr(hi_r), ptr(l8_r), ptr(sc_r), stream_ptr()), "pack")
print("hi equal", torch.equal(hi_p, hi_r), "scales equal", torch.equal(sc_p, sc_r))
got, want = mx.e4m3_decode(l8_p.cpu().numpy()), mx.e4m3_decode(l8_r.cpu().numpy())
x = ps_decode(out, n).cpu().numpy()          # the packed-split kernel's x (hi + lo)
diff = got != want
print("mismatch fraction", diff.mean())
ax = np.abs(x)
for lo_, hi_ in ((0, 1e-6), (1e-6, 1e-4), (1e-4, 1e-2), (1e-2, 1e-1), (1e-1, 1), (1, 1e9)):
    sel = (ax >= lo_) & (ax < hi_)
    if sel.any():
        print(f"|x| in [{lo_:g}, {hi_:g}): {sel.mean():.3f} of values, mismatch {diff[sel].mean():.4f}; negative x share {(x[sel] < 0).mean():.2f}; mismatch among negative {diff[sel & (x < 0)].mean() if (sel & (x < 0)).any() else 0:.4f} positive {diff[sel & (x > 0)].mean() if (sel & (x > 0)).any() else 0:.4f}")
scale = np.repeat(2.0 ** (T._scales(sc_p, m).astype(np.float64) - 127), 32, axis=1)
d_abs = np.abs(got - want) * scale
print("max |lo difference| in x units", d_abs.max(), "relative to |x|:", (d_abs / np.maximum(ax, 1e-30))[diff].max())
# ---- examples: the packed-split kernel's x (exact fp32: hi + lo) beside the MX3 triple's value, for mismatching small negative outputs
hi_v = hi_p.cpu().numpy().view(np.float16)[:, mx.hi_pos(np.arange(n))].astype(np.float64)
x_mx = hi_v + got * scale
idx = np.argwhere(diff & (x < -1e-4) & (x > -1e-2))[:12]
for (r, c) in idx:
    print(f"row {r} col {c}: x_ps {x[r, c]:.9e} ({np.float32(x[r, c]).view(np.uint32):08x}) x_mx {x_mx[r, c]:.9e} hi {hi_v[r, c]:.6e} lo_ps {x[r, c] - hi_v[r, c]:.4e} lo_mx {got[r, c] * scale[r, c]:.4e} want {want[r, c] * scale[r, c]:.4e} scale 2^{int(np.log2(scale[r, c]))}")
