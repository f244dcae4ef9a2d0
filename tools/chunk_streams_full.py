import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from multiplexed_image_annotator_amd import _lib, ops, synth
cells = 49152
dev = _lib.require_gpu()
for name in ("immune_full", "immune_extended"):
    d, c, k = synth.VIT_CONFIGS[name]
    model = ops.VitModel(synth.make_vit_state_dict(name, 1), device=dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    patches = torch.randn((cells, c, 40, 40), generator=g).to(dev)
    src = list(range(c))
    res = {}
    for rnd in range(3):
        for chunk in (1024, 2048, 3072, 4096, 6144):
            for streams in (2, 3):
                model.predict_proba(patches, src, chunk_cells=chunk, streams=streams)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                model.predict_proba(patches, src, chunk_cells=chunk, streams=streams)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((chunk, streams), []).append(cells / e0.elapsed_time(e1))
    for key in sorted(res):
        v = res[key]
        print(f"{name} chunk {key[0]} streams {key[1]}: " + " ".join(f"{x:.2f}" for x in v) + f"  k cells/s (best {max(v):.2f})", flush=True)
