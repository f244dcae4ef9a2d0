"""debug helper: which terms of the MX product does the kernel actually compute?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import mx_emulation as mx
from test_gpu_kernels import ps_decode, ps_encode, rnd
from test_gpu_mx import _ps_halves, _planes
from multiplexed_image_annotator_amd import _lib
from multiplexed_image_annotator_amd._lib import check, lib, ptr, stream_ptr
dev = _lib.require_gpu()
m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
a = rnd((m, k), 5, dev); w = rnd((n, k), 6, dev, 1.0 / np.sqrt(k)); bias = torch.zeros(n, device=dev)
npd = (n + 31) // 32 * 32
z0 = torch.zeros((m, n), device=dev)
a_ps = ps_encode(a, k); w_ps = ps_encode(w, k, (n + 15) // 16 * 16); z_ps = ps_encode(z0, npd)
hi_p, l8_p, sc_p = _planes(m, k, dev)
wh = torch.zeros(lib().ribca_test_mx_weight_bytes(n, k, 0), dtype=torch.uint8, device=dev)
wx = torch.zeros(lib().ribca_test_mx_weight_bytes(n, k, 1), dtype=torch.uint8, device=dev)
check(lib().ribca_test_gemm_mx_resid(ptr(a_ps), 2 * k, ptr(w_ps), 2 * k, m, n, k, ptr(bias), ptr(hi_p), ptr(l8_p), ptr(sc_p), ptr(wh), ptr(wx),
                                     ptr(z_ps), 2 * npd, None, None, None, stream_ptr()), "x")
got = ps_decode(z_ps, n).cpu().numpy()
a_hi, a_lo = _ps_halves(a_ps, k); w_hi, w_lo = _ps_halves(w_ps, k); w_hi, w_lo = w_hi[:n], w_lo[:n]
_, a_lo_q, _, _ = mx.pack_act(a_hi, a_lo.astype(np.float64))
ah, wh_ = a_hi.astype(np.float64), w_hi.astype(np.float64)
T1 = ah @ wh_.T; T2 = a_lo_q @ mx.fp6_image(w_hi).T; T3 = mx.fp6_image(a_hi) @ mx.fp6_image(w_lo).T
exact = a.double().cpu().numpy() @ w.double().cpu().numpy().T
sc = 1.0 + np.abs(a.cpu().numpy().astype(np.float64)) @ np.abs(w.cpu().numpy().astype(np.float64)).T
def e(x): return (np.abs(got - x) / sc).max()
print("T1 only", e(T1)); print("T1+T2", e(T1 + T2)); print("T1+T3", e(T1 + T3)); print("T1+T2+T3", e(T1 + T2 + T3)); print("exact", e(exact))
for f2 in (0.5, 2.0, -1.0):
    print("T1 + %g T2 + T3" % f2, e(T1 + f2 * T2 + T3)); print("T1 + T2 + %g T3" % f2, e(T1 + T2 + f2 * T3))
# per column-tile error pattern
d = np.abs(got - (T1 + T2 + T3)) / sc
print("err by 16-col tile:", [float("%.2e" % d[:, j:j + 16].max()) for j in range(0, n, 16)][:24])
print("err by row tile:", [float("%.2e" % d[i:i + 16].max()) for i in range(0, m, 16)][:10])
r = got - T1
print("corr(got-T1, T2)", np.sum(r * T2) / np.sum(T2 * T2), " corr(got-T1, T3)", np.sum(r * T3) / np.sum(T3 * T3))
