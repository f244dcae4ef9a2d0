#!/usr/bin/env python3
"""(GPU box) how well the 64-cell load-time probe of ops.VitModel predicts the largest |fast (MX) - full precision| over real cells: both weight
families, several seeds, the four classifiers that can use the MX products, 6000 cells of BASELINE config 3's patches (GPU only: no oracle).
The rule of DESIGN 3.5 rests on the ratio printed here: predicted worst |dp| = 3 x delta / 4 with delta the probe's largest move of a logit
difference; accepted while <= RECHECK_MARGIN / 2.5.
usage: python tools/probe_vs_real.py [cells] [seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from multiplexed_image_annotator_amd import _lib, ops, synth
import test_gpu_e2e as T
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = _lib.require_gpu()
seed0, mask, image, ids, tab, cmin = T._config3_inputs(dev)
sel = np.linspace(0, len(ids) - 1, n_cells).astype(np.int64)
patches, _ = ops.extract_patches(image, mask, cmin, torch.from_numpy(ids[sel].astype(np.int32)).to(dev), torch.from_numpy(tab[sel, :4].astype(np.int32)).to(dev))
worst = 0.0
for family, kw in (("uniform head 4", {}), ("uniform head 1.5", {"head_gain": 1.5}), ("heavy head 4", {}), ("heavy head 1.5", {"head_gain": 1.5})):
    make = synth.make_vit_state_dict_heavy if family.startswith("heavy") else synth.make_vit_state_dict
    for s in range(n_seeds):
        for name in ("immune_base", "struct", "immune_extended", "immune_full"):
            d, c, k = synth.VIT_CONFIGS[name]
            vm = ops.VitModel(make(name, seed0 + 101 * s, **kw), dev)
            fast = vm._forward(patches, list(range(c)), 1024, 0, 1, precise=False, force_fast=True)
            full = vm._forward(patches, list(range(c)), 1024, 0, 1, precise=True)
            real = float((fast - full).abs().max())
            ratio = real / max(vm.probe_fast_minus_full, 1e-12)
            def logit_delta(a, b):
                a, b = a.double(), b.double()
                ok = (a > 1e-30) & (b > 1e-30)
                la, lb = torch.log(a.clamp_min(1e-300)), torch.log(b.clamp_min(1e-300))
                top = b.argmax(1, keepdim=True)
                d = ((la - la.gather(1, top)) - (lb - lb.gather(1, top))).abs()
                return float(d[ok].max())
            dl_real = logit_delta(fast, full)
            if vm.fast_ok:
                worst = max(worst, real)
            print(f"{family:18s} seed+{101 * s:<4d} {name:16s} probe: logit-difference move {vm.probe_logit_delta:.2e} -> predicted worst |dp| {vm.probe_predicted_dp:.2e} "
                  f"({'accepted' if vm.fast_ok else 'REFUSED '}) | {n_cells} real cells: |dp| max {real:.2e} = {real / max(0.25 * vm.probe_logit_delta, 1e-12):.2f} x delta / 4, "
                  f"logit-difference move {dl_real:.2e} | probability form of the probe {vm.probe_fast_minus_full:.2e} (real / it = {ratio:.1f})", flush=True)
            del vm, fast, full
print(f"largest real |fast - full| among ACCEPTED models: {worst:.2e} (margin / 2.5 = 4.0e-04)")
