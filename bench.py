#!/usr/bin/env python3
"""Headline benchmark: cells/sec annotated on a synthetic 15-channel 4096x4096 tile with 100k cells, Full Panel,
all five ViT classifiers per cell (BASELINE.json configs[2]; with --gpus N>1 the same tile's cells are sharded across
ranks with one RCCL all-gather of the per-cell probabilities = configs[3]).

One step = one pass of the hot path over the tile, inputs (uint16 image, int32 mask, packed weights) resident in HBM:
normalise -> label table -> per-cell crop / soft mask -> 5 x (patch-embed, 12 blocks, head, softmax) -> vote -> labels
on the host.  Prints ONE JSON line (rank 0).  ``roofline`` is measured live with HIP events around every GEMM launch of
one extra (untimed) profiled pass; ``cpu_baseline`` times the CPU oracle on a bounded sample of the same workload;
``dropin`` times the boundary itself (``Annotator.preprocess -> predict -> export_annotations`` from host ``.npy`` files,
H2D copies and the CSV included) on the same tile.

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (``torch.distributed.run`` as a child
process, before this process touches a GPU); under the driver's own ``torch.distributed.run`` launch it is just a rank.
``--impute`` switches the workload to BASELINE config 5 (one full-panel marker missing -> MAE imputer + the five ViTs).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_DENSE_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=100000)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--channels", type=int, default=15)
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("RIBCA_CHUNK_CELLS", "1024")))
    ap.add_argument("--streams", type=int, default=int(os.environ.get("RIBCA_STREAMS", "3")), help="each classifier's cells are split into this many segments enqueued on separate HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the Annotator (boundary path) timing")
    ap.add_argument("--impute", action="store_true", help="BASELINE config 5: last full-panel marker missing, imputed by the MAE (infer=True)")
    ap.add_argument("--cpu-sample", type=int, default=512, help="cells of the CPU-oracle sample")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous + one all-gather only (no GPU work): CPU test of the N-rank launch path")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """Start ``args.gpus`` ranks of this script under torch.distributed.run as a CHILD process.  Nothing in this (parent) process
    has touched a GPU: counting devices does not initialise HIP, and the library is only compiled here, not loaded."""
    import socket
    import subprocess
    share = os.environ.get("RIBCA_SHARE_GPU") == "1" or args.launch_check
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    if not args.launch_check:
        from multiplexed_image_annotator_amd import build as _build
        _build.build(verbose=False)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def launch_check(args, world, rank, backend):
    """The N-rank plumbing without a GPU: process group, barrier, the padded all-gather of dist.all_gather_rows, JSON on rank 0."""
    import torch.distributed as tdist
    from multiplexed_image_annotator_amd import dist
    if world > 1:
        tdist.init_process_group("gloo" if backend != "nccl" or not torch.cuda.is_available() else backend)
    n = 1001
    lo, hi = dist.shard_bounds(n, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1).repeat(1, 33)
    full = dist.all_gather_rows(local, n)
    ok = bool(torch.equal(full[:, 0], torch.arange(n, dtype=torch.float32)))
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "launch check (no measurement)", "value": None, "n_gpus": world, "gather_ok": ok}))
    return 0 if ok else 1


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RIBCA_DIST_BACKEND", "nccl")     # "gloo" + RIBCA_SHARE_GPU=1: rehearsal of the N > 1 path on a 1-GPU box
    if args.launch_check:
        sys.exit(launch_check(args, world, rank, backend))
    if world > 1:
        import torch.distributed as tdist
        if os.environ.get("RIBCA_SHARE_GPU") == "1":
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            tdist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            tdist.init_process_group(backend)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        tdist.barrier()
    from multiplexed_image_annotator_amd import _lib, dist, ops, synth
    from multiplexed_image_annotator_amd.annotator import CLASS_NAMES
    from multiplexed_image_annotator_amd.marker_parse import PANELS
    dev = _lib.require_gpu()
    seed = synth.SEED_BASE + 3

    # ---- synthetic inputs, generated on the device (bit-identical on every rank) ------------------------------------
    mask, img = synth.make_mask_and_image(args.size, args.size, args.cells, args.channels, seed, device=dev)
    raw = img.to(torch.int16)            # uint16 bit pattern (values < 65536)
    del img
    markers = synth.FULL_PANEL_MARKERS[:args.channels]
    models, srcs = {}, {}
    for name, (d, c, k) in synth.VIT_CONFIGS.items():
        if c > args.channels:
            continue
        models[name] = ops.VitModel(synth.make_vit_state_dict(name, seed), dev)
        panel = {"immune_base": "immune_base", "immune_extended": "immune_extended", "immune_full": "immune_full"}.get(name)
        if panel and all(m in markers for m in PANELS[panel]):
            srcs[name] = [markers.index(m) for m in PANELS[panel]]
        else:
            srcs[name] = list(range(c))  # struct / nerve markers are not in a 15-marker immune panel: fixed synthetic mapping
    flops_cell = sum(m.flops_per_cell for m in models.values())
    gid = {n: i for i, n in enumerate(ops.GLOBAL_NAMES)}
    tc = [-1.0] * 18
    vote_pair = ("immune_full", "struct") if "immune_full" in models and "struct" in models else (next(iter(models)), None)

    def one_pass(streams=None):
        streams = args.streams if streams is None else streams
        image = ops.normalize_image(raw, blur=0.3, amax=99.8, u16_bits=True)
        ids, tab = ops.label_table(mask)
        n = len(ids)
        lo, hi = dist.shard_bounds(n, rank, world)
        cmin = ops.channel_min(image)
        ids_d = torch.from_numpy(ids[lo:hi].astype(np.int32)).to(dev)
        bb_d = torch.from_numpy(tab[lo:hi, :4].astype(np.int32)).to(dev)
        patches, _ = ops.extract_patches(image, mask, cmin, ids_d, bb_d)
        probs = {}
        for name, model in models.items():
            probs[name] = model.predict_proba(patches, srcs[name], chunk_cells=args.chunk, streams=streams)
        if world > 1:       # ONE all-gather per tile: the five models' probability columns side by side (33 floats per cell)
            names = list(probs)
            widths = [probs[k].shape[1] for k in names]
            full = dist.all_gather_rows(torch.cat([probs[k] for k in names], dim=1), n)
            probs = {k: t.contiguous() for k, t in zip(names, torch.split(full, widths, dim=1))}
        a, b = vote_pair
        lab, conf = ops.vote(probs[a], [gid[c] for c in CLASS_NAMES[a]], probs[b] if b else None,
                             [gid[c] for c in CLASS_NAMES[b]] if b else None, tc, 0.3)
        return n, lab.cpu(), conf.cpu()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"inputs ready: {len(models)} models, {flops_cell / 1e9:.3f} GFLOP/cell")
    for i in range(args.warmup):
        one_pass()
        note(f"warmup {i + 1}/{args.warmup} done")
    sync_all()
    t0 = time.perf_counter()
    n_cells = 0
    for i in range(args.steps):
        n_cells, lab, conf = one_pass()
        note(f"step {i + 1}/{args.steps} done at {time.perf_counter() - t0:.2f}s")
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = n_cells * args.steps / dt

    out = {
        "metric": "cells/sec annotated (whole node)", "value": round(value, 2), "unit": "cells/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"synthetic {args.channels}-ch {args.size}x{args.size} tile, {n_cells} cells, Full Panel, "
                               f"{len(models)} ViT classifiers per cell (normalise + label table + crop/soft-mask + ViT + vote)",
                   "cells": n_cells, "models": list(models), "chunk_cells": args.chunk, "segment_streams": args.streams, "precision": "bf16x3 split MFMA, fp32 accumulate",
                   "parallelism": f"cells sharded over {world} rank(s), all-gather of per-cell probabilities" if world > 1 else "single GPU"},
        "vit_gflop_per_cell": round(flops_cell / 1e9, 4),
        "vit_mfma_util_vs_bf16_dense": round(value * flops_cell / (world * PEAK_BF16_DENSE_TFLOPS * 1e12), 5),
    }

    # ---- roofline of the dominant kernel (the bf16x3 GEMM family), one extra profiled pass -----------------------------
    if not args.no_roofline and rank == 0:
        ops.prof_enable(True)
        one_pass(streams=1)      # per-kernel durations: one stream, so a launch's events bracket that launch alone
        torch.cuda.synchronize()
        prof = ops.prof_read()
        ops.prof_enable(False)
        lo, hi = dist.shard_bounds(n_cells, rank, world)
        n_local = hi - lo
        gemm_flops = 0.0
        for name, model in models.items():
            d = model.D
            # qkv + proj + fc1 + fc2 (algorithmic, unpadded); in the last block proj / fc1 / fc2 run on the CLS row only
            gemm_flops += n_local * ((model.depth - 1) * 24.0 * 101 * d * d + 6.0 * 101 * d * d + 18.0 * d * d)
        g_ms = sum(prof[k][0] for k in ("gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2"))
        g_n = sum(prof[k][1] for k in ("gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2"))
        achieved = gemm_flops / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_final", "gemm_traffic.json")
        if os.path.exists(tpath):                      # HBM/fabric bytes per launch from the committed rocprofv3 PMC passes
            traffic = round(json.load(open(tpath))["traffic_bytes_per_launch"])
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_ps_split_kernel (qkv/proj/fc1/fc2, bf16x3)", "achieved": round(achieved, 2),
                           "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 5),
                           "traffic": traffic, "traffic_unit": "bytes/launch (2*FETCH_SIZE + WRITE_SIZE, profiles/r1_final/gemm_traffic.json)",
                           "launches": int(g_n), "avg_launch_ms": round(g_ms / max(g_n, 1), 5),
                           "algorithmic_gflop_per_launch": round(gemm_flops / max(g_n, 1) / 1e9, 4), "mfma_passes_per_product": 3,
                           "per_kernel_ms": {k: round(v[0], 3) for k, v in prof.items() if v[1]}}

    # ---- CPU baseline: the oracle (numpy/scipy/torch fp32 restatement of the reference path) on a bounded sample ------------
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, raw, mask, markers, seed, n_cells)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        tdist.destroy_process_group()


def cpu_baseline(args, raw_dev, mask_dev, markers, seed, n_cells):
    """kind 'port': oracle/ on the host cores.  Sample: normalise + label scan on a 1024x1024 corner of the same tile (scaled by
    area), crop/soft-mask + the same five ViTs on 48 of its cells; per-cell times are combined into end-to-end cells/s."""
    from multiplexed_image_annotator_amd import synth
    from oracle import ref_preprocess as rp, ref_vit
    threads = min(16, len(os.sched_getaffinity(0)))      # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(threads)
    side = min(1024, args.size)
    raw = raw_dev[:, :side, :side].cpu().numpy().view(np.uint16)
    mask = mask_dev[:side, :side].cpu().numpy()
    area_scale = (args.size * args.size) / float(side * side)
    t = time.perf_counter()
    image = rp.normalize_image(raw, blur=0.3, amax=99.8)
    t_norm = (time.perf_counter() - t) * area_scale
    t = time.perf_counter()
    ids, tab = rp.cell_table(mask)
    t_label = (time.perf_counter() - t) * area_scale
    k = min(48, len(ids))
    sel = np.linspace(0, len(ids) - 1, k).astype(int)
    t = time.perf_counter()
    patches, _ = rp.patches_for_panel(image, mask, list(range(raw.shape[0])), ids[sel], tab[sel], want_intensity=False)
    t_crop = (time.perf_counter() - t) / k
    x = torch.from_numpy(patches)
    t_vit = 0.0
    n_models = 0
    for name, (d, c, kk) in synth.VIT_CONFIGS.items():
        if c > raw.shape[0]:
            continue
        sd = synth.make_vit_state_dict(name, seed)
        t = time.perf_counter()
        ref_vit.predict_proba(sd, x[:, :c], 48)
        t_vit += (time.perf_counter() - t) / k
        n_models += 1
    per_cell = (t_norm + t_label) / n_cells + n_models * t_crop + t_vit     # the reference crops once per applicable panel
    return {"value": round(1.0 / per_cell, 3), "unit": "cells/s", "cores": threads, "kind": "port",
            "sample": f"oracle on a {side}x{side} corner: normalise+label scan scaled by area to the full tile ({t_norm:.1f}s+{t_label:.2f}s), "
                      f"crop/soft-mask {t_crop*1e3:.2f} ms/cell/panel (1 thread, as the reference) and {n_models} fp32 ViTs "
                      f"{t_vit*1e3:.1f} ms/cell ({threads} threads, batch 48) on {k} cells"}


if __name__ == "__main__":
    main()
