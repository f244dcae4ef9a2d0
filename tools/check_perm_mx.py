"""permutation / repeatability check of the classifiers on real-shaped patches (debug helper; mirrors tests/test_gpu_e2e.py::test_full_panel_properties)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from multiplexed_image_annotator_amd import _lib, ops, synth
dev = _lib.require_gpu()
names = sys.argv[1:] or ["immune_base", "immune_extended", "immune_full"]
seed = synth.SEED_BASE + 3
mask, img = synth.make_mask_and_image(1536, 1536, 12000, 15, seed, device=dev)
image = ops.normalize_image(img.to(torch.float32), blur=0.3, amax=99.8)
ids, tab = ops.label_table(mask)
n = len(ids)
cmin = ops.channel_min(image)
ids_d = torch.from_numpy(ids.astype(np.int32)).to(dev); bb_d = torch.from_numpy(tab[:, :4].astype(np.int32)).to(dev)
patches, _ = ops.extract_patches(image, mask, cmin, ids_d, bb_d)
lo, hi = n // 3, n // 3 + 777
for name in names:
    d, c, k = synth.VIT_CONFIGS[name]
    model = ops.VitModel(synth.make_vit_state_dict(name, seed, depth=int(os.environ.get("DEPTH", "12"))), dev)
    src = list(range(c))
    full = model.predict_proba(patches, src, chunk_cells=256)
    a = model.predict_proba(patches[lo:hi].contiguous(), src, chunk_cells=100)
    a2 = model.predict_proba(patches[lo:hi].contiguous(), src, chunk_cells=100)
    perm = torch.randperm(hi - lo, generator=torch.Generator().manual_seed(1)).to(dev)
    b = model.predict_proba(patches[lo:hi][perm].contiguous(), src, chunk_cells=64)
    b2 = model.predict_proba(patches[lo:hi][perm].contiguous(), src, chunk_cells=64)
    diff = (b - a[perm]).abs().max(1).values
    print(name, "shard==slice:", bool(torch.equal(a, full[lo:hi])), "repeat:", bool(torch.equal(a, a2)), bool(torch.equal(b, b2)), "| perm equal:", bool(torch.equal(b, a[perm])),
          "| rows differing:", int((diff > 0).sum().item()), "max diff %.3e" % diff.max().item(), "| first positions:", torch.nonzero(diff > 0).flatten()[:16].tolist(), flush=True)
