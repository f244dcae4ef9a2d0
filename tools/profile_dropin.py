#!/usr/bin/env python3
"""Host-side profile (cProfile) of the drop-in path on the config-3 tile: where the Annotator spends time outside the kernels."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import __graft_entry__
__graft_entry__.build()
from multiplexed_image_annotator_amd import _lib, ops, synth
from multiplexed_image_annotator_amd.annotator import Annotator

dev = _lib.require_gpu()
seed = synth.SEED_BASE + 3
mask, img = synth.make_mask_and_image(4096, 4096, 100000, 15, seed, device=dev)
tmp = tempfile.mkdtemp(prefix="ribca_prof_")
np.save(os.path.join(tmp, "img.npy"), img.to(torch.int16).cpu().numpy().view(np.uint16))
np.save(os.path.join(tmp, "mask.npy"), mask.to(torch.int32).cpu().numpy())
del img, mask
open(os.path.join(tmp, "markers.txt"), "w").write("\n".join(synth.FULL_PANEL_MARKERS) + "\n")
open(os.path.join(tmp, "images.csv"), "w").write("image_path,mask_path\n%s,%s\n" % (os.path.join(tmp, "img.npy"), os.path.join(tmp, "mask.npy")))
model = ops.VitModel(synth.make_vit_state_dict("immune_full", seed), dev)


def run_once():
    t = [time.perf_counter()]
    a = Annotator(os.path.join(tmp, "markers.txt"), os.path.join(tmp, "images.csv"), "cuda", tmp, "p", True, False, -1, True, 0.3, 99.8, 0.3, 30, None)
    a.models, a._loaded = {"immune_full": model}, True
    t.append(time.perf_counter())
    a.preprocess(); torch.cuda.synchronize()
    t.append(time.perf_counter())
    a.predict(128); torch.cuda.synchronize()
    t.append(time.perf_counter())
    a.export_annotations()
    t.append(time.perf_counter())
    a.clear_tmp(); a.logger.close()
    return [round((b - a_) * 1e3, 1) for a_, b in zip(t[:-1], t[1:])]


print("init / preprocess / predict / export (ms):", run_once())
print("init / preprocess / predict / export (ms):", run_once())
pr = cProfile.Profile()
pr.enable()
run_once()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
