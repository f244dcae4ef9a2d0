#!/usr/bin/env python3
"""Do two classifiers of different width run faster side by side (each on its own streams) than one after the other?  The bench runs the
five classifiers one after the other, each split into three segment streams.  usage: python tools/concurrent_models.py [cells]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multiplexed_image_annotator_amd import _lib, ops, synth

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
dev = _lib.require_gpu()
names = ["immune_full", "immune_extended", "immune_base", "nerve"]
models, patches = {}, {}
g = torch.Generator(device="cpu").manual_seed(0)
for n in names:
    d, c, k = synth.VIT_CONFIGS[n]
    models[n] = ops.VitModel(synth.make_vit_state_dict(n, 1), device=dev)
    patches[n] = torch.randn((cells, c, 40, 40), generator=g).to(dev)


def run_seq(sel, streams=3):
    for n in sel:
        models[n].predict_proba(patches[n], list(range(models[n].C)), streams=streams, ws_slot=0)


def run_par(sel, streams=2):
    main = torch.cuda.current_stream()
    outer = [torch.cuda.Stream() for _ in sel]
    for i, n in enumerate(sel):
        outer[i].wait_stream(main)
        with torch.cuda.stream(outer[i]):
            models[n].predict_proba(patches[n], list(range(models[n].C)), streams=streams, ws_slot=1 + i)
    for st in outer:
        main.wait_stream(st)


def timed(fn, *a, **kw):
    fn(*a, **kw); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(*a, **kw); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for sel in (["immune_full", "nerve"], ["immune_full", "immune_base"], ["immune_full", "immune_extended"], names):
    s3 = timed(run_seq, sel, 3)
    p1 = timed(run_par, sel, 1)
    print(f"{' + '.join(sel)} ({cells} cells each): one after the other (3 segment streams each) {s3:.1f} ms | side by side, 1 stream each {p1:.1f} ms "
          f"({100 * (p1 / s3 - 1):+.1f} %)", flush=True)
