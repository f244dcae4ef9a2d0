#!/usr/bin/env python3
"""(GPU box) heavy-tailed weight family on BASELINE config 3's real patches: this path (fast / full precision) against the fp32 AND the fp64 CPU
oracle -- how much of |this path - fp32 reference| is the reference's own distance from exact arithmetic.
usage: python tools/heavy_probe.py [cells] [family]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from multiplexed_image_annotator_amd import _lib, ops, synth
from oracle import ref_vit
import test_gpu_e2e as T
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 300
family = sys.argv[2] if len(sys.argv) > 2 else "heavy"
dev = _lib.require_gpu()
seed, mask, image, ids, tab, cmin = T._config3_inputs(dev)
sel = np.linspace(0, len(ids) - 1, n_cells).astype(np.int64)
patches, _ = ops.extract_patches(image, mask, cmin, torch.from_numpy(ids[sel].astype(np.int32)).to(dev), torch.from_numpy(tab[sel, :4].astype(np.int32)).to(dev))
x = patches.cpu()
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
for name, (d, c, k) in synth.VIT_CONFIGS.items():
    sd = synth.WEIGHT_FAMILIES[family](name, seed, head_gain=1.5)
    with torch.no_grad():
        feat = torch.cat([ref_vit.forward_features(sd, x[i:i + 100, :c]) for i in range(0, n_cells, 100)])
        sd["head.bias"] = synth.calibrate_head_bias(sd, feat[:256])
        p32 = torch.softmax(torch.nn.functional.linear(feat, sd["head.weight"], sd["head.bias"]), dim=1).double()
        sd64 = {kk: v.double() for kk, v in sd.items()}
        p64 = torch.cat([torch.softmax(ref_vit.logits(sd64, x[i:i + 100, :c].double()), dim=1) for i in range(0, n_cells, 100)])
    vm = ops.VitModel(sd, dev)
    fast = vm._forward(patches, list(range(c)), 1024, 0, 1, precise=False).cpu().double()
    full = vm._forward(patches, list(range(c)), 1024, 0, 1, precise=True).cpu().double()
    f = lambda a, b: float((a - b).abs().max())
    print(f"{family} {name}: probe delta {vm.probe_logit_delta:.1e} | fp32 ref vs fp64 {f(p32, p64):.2e} | fast vs fp32 {f(fast, p32):.2e} vs fp64 {f(fast, p64):.2e} | "
          f"full vs fp32 {f(full, p32):.2e} vs fp64 {f(full, p64):.2e} | fast vs full {f(fast, full):.2e}", flush=True)
