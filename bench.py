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
    ap.add_argument("--cpu-sample", type=int, default=2000, help="cells of the CPU-oracle sample (SURVEY 8(d): 2000)")
    ap.add_argument("--config1", action="store_true", help="BASELINE configs[0] stand-in (example_1 mask, Basic panel, bs 8): drop-in on the GPU "
                                                            "beside the CPU oracle timed IN FULL")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous + one all-gather only (no GPU work): CPU test of the N-rank launch path")
    return ap.parse_args()


def visible_gpu_count() -> int:
    """GPUs this process tree will see, WITHOUT any HIP / torch.cuda call (the launcher parent must never initialise the GPU:
    it starts the ranks as children): HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES if set, else the
    kfd topology (nodes with simd_count > 0 and a gfx target are GPUs)."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    base = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        for node in sorted(os.listdir(base)):
            try:
                props = dict(l.split(None, 1) for l in open(os.path.join(base, node, "properties")).read().splitlines() if " " in l)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0 and int(props.get("gfx_target_version", "0")) > 0:
                n += 1
    except OSError:
        return 0
    return n


def launch_ranks(args) -> int:
    """Start ``args.gpus`` ranks of this script under torch.distributed.run as a CHILD process.  Nothing in this (parent) process
    touches a GPU: devices are counted from the environment / sysfs (no HIP runtime call), the library is only compiled here."""
    import socket
    import subprocess
    share = os.environ.get("RIBCA_SHARE_GPU") == "1" or args.launch_check
    have = visible_gpu_count()
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
    t0 = time.perf_counter()
    full = dist.all_gather_rows(local, n)
    ag_ms = (time.perf_counter() - t0) * 1e3
    ok = bool(torch.equal(full[:, 0], torch.arange(n, dtype=torch.float32)))
    coll = collective_record(world, ag_ms, 1, local)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "launch check (no measurement)", "value": None, "n_gpus": world, "gather_ok": ok, "collective": coll,
                          "visible_gpus_no_hip": visible_gpu_count()}))
    return 0 if ok else 1


def collective_record(world, allgather_ms_total, steps, local_rows):
    """What the data-path collective actually was, so that an N > 1 line can be checked: the backend torch.distributed reports
    ("nccl" = RCCL on ROCm), the world size THE PROCESS GROUP sees, the time of the per-tile all-gather and its payload."""
    import torch.distributed as tdist
    if world <= 1 or not tdist.is_initialized():
        return {"backend": None, "world_size_seen": 1, "allgather_ms_per_step": 0.0, "bytes_per_rank": 0}
    return {"backend": tdist.get_backend(), "world_size_seen": tdist.get_world_size(),
            "allgather_ms_per_step": round(allgather_ms_total / max(steps, 1), 4),
            "bytes_per_rank": int(local_rows.numel() * local_rows.element_size()),
            "call": "ONE all_gather_into_tensor per tile (dist.all_gather_rows), shards padded to the largest"}


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
    if args.config1:
        import contextlib
        import io
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):             # the Annotator's own prints go to stderr, the JSON line alone to stdout
            rc = config1_bench(args)
        lines = buf.getvalue().splitlines()
        print("\n".join(l for l in lines if not l.startswith("{")), file=sys.stderr)
        for l in lines:
            if l.startswith("{"):
                print(l)
        sys.exit(rc)
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
    # config 3/4: one tile, seed base + 3, cells sharded over the ranks; config 5 (--impute): one tile PER RANK (seed base + 5 + rank),
    # replicas only -- no data-path collective
    seed = synth.SEED_BASE + (5 + rank if args.impute else 3)
    sharded = world > 1 and not args.impute

    # ---- synthetic inputs, generated on the device (bit-identical on every rank) ------------------------------------
    mask, img = synth.make_mask_and_image(args.size, args.size, args.cells, args.channels, seed, device=dev)
    raw = img.to(torch.int16)            # uint16 bit pattern (values < 65536)
    del img
    markers = list(synth.FULL_PANEL_MARKERS[:args.channels])
    imputer, imp_present = None, None
    if args.impute:
        # BASELINE config 5: the last full-panel marker ('Trypase') is replaced by a marker outside the panel,
        # so the full panel's index list is [0..13, -1] and the MAE imputer fills the missing plane of every cell (infer=True)
        if args.channels != 15:
            raise SystemExit("--impute needs the 15-channel full panel")
        markers[-1] = "CollagenIV"
        imputer = ops.MaeModel(synth.make_mae_state_dict("immune_full", seed), dev)
        imp_present = list(range(14))
    models, srcs = {}, {}
    for name, (d, c, k) in synth.VIT_CONFIGS.items():
        if c > args.channels:
            continue
        models[name] = ops.VitModel(synth.make_vit_state_dict(name, seed), dev)
        panel = {"immune_base": "immune_base", "immune_extended": "immune_extended", "immune_full": "immune_full"}.get(name)
        if panel and all(m in markers for m in PANELS[panel]):      # (with --impute the full panel misses one marker: handled in one_pass)
            srcs[name] = [markers.index(m) for m in PANELS[panel]]
        else:
            srcs[name] = list(range(c))  # struct / nerve markers are not in a 15-marker immune panel: fixed synthetic mapping
    flops_cell = sum(m.flops_per_cell for m in models.values())
    gid = {n: i for i, n in enumerate(ops.GLOBAL_NAMES)}
    tc = [-1.0] * 18
    vote_pair = ("immune_full", "struct") if "immune_full" in models and "struct" in models else (next(iter(models)), None)

    ag_events, ag_state = [], {}
    norm_shard = os.environ.get("RIBCA_NORM_SHARD") == "1"      # cell-sharded runs only; default: every rank normalises the whole tile

    # as Annotator.predict does: cells whose fast (MX) result lies within 1e-3 of a decision boundary (top-2 margin, the vote's confidence
    # threshold) are re-evaluated with three fp16 passes per product INSIDE the timed region
    RECHECK = [0.3]
    recheck_counts = {}     # last timed pass: the vote pair's cells re-evaluated / left inside the noise floor (Annotator.predict's own criterion)
    stage_events = []      # per timed pass: [(stage, start event, end event)] on the current stream (the ViT's segment streams join it)
    STAGES = ("normalise", "label_table", "crop", "imputer", "vit", "all_gather", "vote", "d2h")

    def one_pass(streams=None, models_sel=None, record=False, gather=True):
        # gather=False: no collective (the profiled per-classifier passes of the roofline block run on rank 0 ALONE -- a collective there
        # would wait for ranks that have already left); the vote then runs on this rank's shard
        streams = args.streams if streams is None else streams
        marks = []

        def stage(name):      # context manager: two events around a stage, read after the timed region (no synchronisation here)
            class _S:
                def __enter__(self_inner):
                    if record:
                        self_inner.a = torch.cuda.Event(enable_timing=True); self_inner.a.record()
                def __exit__(self_inner, *exc):
                    if record:
                        b = torch.cuda.Event(enable_timing=True); b.record()
                        marks.append((name, self_inner.a, b))
                    return False
            return _S()

        with stage("normalise"):
            if sharded and norm_shard:
                # channel-sharded form (Annotator: RIBCA_NORM_SHARD=1): this rank's ceil(C / world) channels, one all-gather of the fp32 planes
                c0, c1 = dist.shard_bounds(raw.shape[0], rank, world)
                local = (ops.normalize_image(raw[c0:c1], blur=0.3, amax=99.8, u16_bits=True) if c1 > c0
                         else torch.empty((0,) + tuple(raw.shape[1:]), dtype=torch.float32, device=dev))
                image = dist.all_gather_planes(local, raw.shape[0])
            else:
                image = ops.normalize_image(raw, blur=0.3, amax=99.8, u16_bits=True)
        with stage("label_table"):
            # ids and boxes stay on the device: the crop reads them there (only the label range and the cell count cross PCIe)
            ids_all, bb_all = ops.label_table_device(mask)
            n = int(ids_all.numel())
        lo, hi = dist.shard_bounds(n, rank, world) if sharded else (0, n)
        with stage("crop"):
            cmin = ops.channel_min(image)
            ids_d = ids_all[lo:hi].contiguous()
            bb_d = bb_all[lo:hi].contiguous()
            patches, _ = ops.extract_patches(image, mask, cmin, ids_d, bb_d)
        probs = {}
        imputed_panel = None
        for name, model in models.items():
            if models_sel is not None and name not in models_sel:
                continue
            if imputer is not None and name == "immune_full":
                # reference preprocess.py:268-281: the panel tensor with its missing plane imputed feeds that panel's classifier
                with stage("imputer"):
                    panel = patches[:, :15].contiguous()
                    imputer.impute(panel, imp_present, chunk_cells=ops.MaeModel.CHUNK_FACTOR * args.chunk)
                with stage("vit"):
                    # (a member of the vote pair: re-evaluated below by the vote's own distance -- on the IMPUTED panel, kept for that)
                    probs[name] = model.predict_proba(panel, list(range(15)), chunk_cells=args.chunk, streams=streams,
                                                      recheck=None if (models_sel is None and name in vote_pair) else RECHECK)
                imputed_panel = panel
                del panel
                continue
            with stage("vit"):
                # the two classifiers the vote reads are re-evaluated below by the vote's own distance (ops.decision_distance, what
                # Annotator._recheck_near_boundaries does); the others keep the per-classifier rule (top-2 margin + the confidence threshold)
                pair_member = models_sel is None and name in vote_pair
                probs[name] = model.predict_proba(patches, srcs[name], chunk_cells=args.chunk, streams=streams, recheck=None if pair_member else RECHECK)
        if models_sel is None and vote_pair[1] is not None:
            with stage("vit"):
                a_, b_ = vote_pair
                oth = {k: (CLASS_NAMES[k].index("Others") if "Others" in CLASS_NAMES[k] else None) for k in vote_pair}
                dd = ops.decision_distance(probs[a_], oth[a_], probs[b_], oth[b_], RECHECK)
                uses_mx = [k for k in vote_pair if models[k].uses_mx]
                rows = torch.nonzero(dd < max([models[k].recheck_margin for k in uses_mx] or [ops.VitModel.RECHECK_MARGIN])).flatten()
                if rows.numel() and uses_mx:
                    for k in uses_mx:
                        from_imputed = imputed_panel is not None and k == "immune_full"
                        src_p = imputed_panel if from_imputed else patches
                        probs[k].index_copy_(0, rows, models[k]._forward(src_p.index_select(0, rows), list(range(15)) if from_imputed else srcs[k],
                                                                         args.chunk, 0, 1, precise=True))
                    dd = ops.decision_distance(probs[a_], oth[a_], probs[b_], oth[b_], RECHECK)
                if record:
                    recheck_counts["vote_pair_cells_re_evaluated"] = int(rows.numel()) if uses_mx else 0
                    recheck_counts["vote_pair_cells_within_noise_floor"] = int((dd < 2.0e-4).sum().item())
        if sharded and gather:         # ONE all-gather per tile: the five models' probability columns side by side (33 floats per cell)
            names = list(probs)
            widths = [probs[k].shape[1] for k in names]
            local_rows = torch.cat([probs[k] for k in names], dim=1)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            full = dist.all_gather_rows(local_rows, n)
            ev1.record()
            ag_events.append((ev0, ev1))
            if record:
                marks.append(("all_gather", ev0, ev1))
            ag_state["local"] = local_rows
            probs = {k: t.contiguous() for k, t in zip(names, torch.split(full, widths, dim=1))}
        a, b = vote_pair if models_sel is None else (next(iter(probs)), None)
        with stage("vote"):
            lab, conf = ops.vote(probs[a], [gid[c] for c in CLASS_NAMES[a]], probs[b] if b else None,
                                 [gid[c] for c in CLASS_NAMES[b]] if b else None, tc, 0.3)
        with stage("d2h"):
            lab_h, conf_h = lab.cpu(), conf.cpu()
        if record:
            stage_events.append(marks)
        return n, lab_h, conf_h

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
    ag_events.clear()
    t0 = time.perf_counter()
    n_cells = 0
    for i in range(args.steps):
        n_cells, lab, conf = one_pass(record=True)
        note(f"step {i + 1}/{args.steps} done at {time.perf_counter() - t0:.2f}s")
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    total_cells = n_cells
    if world > 1 and args.impute:       # replicas: every rank annotated its own tile
        t = torch.tensor([n_cells], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
        total_cells = int(t.item())
    value = total_cells * args.steps / dt

    out = {
        "metric": "cells/sec annotated (whole node)", "value": round(value, 2), "unit": "cells/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak" if args.impute else "strong", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"synthetic {args.channels}-ch {args.size}x{args.size} tile, {n_cells} cells, Full Panel, "
                               + ("one marker missing -> MAE imputer (infer=True) + " if args.impute else "")
                               + f"{len(models)} ViT classifiers per cell (normalise + label table + crop/soft-mask + ViT + vote)",
                   "baseline_config": "configs[4] (one tile per GPU, imputation)" if args.impute else ("configs[2]" if world == 1 else "configs[3]"),
                   "cells": n_cells, "models": list(models), "chunk_cells": args.chunk,
                   "chunk_cells_effective": {name: m.effective_chunk(args.chunk) for name, m in models.items()}, "segment_streams": args.streams,
                   "precision": "fp16 hi+lo split operands, fp32 accumulate: 3 fp16 MFMA passes per product, or (mlp.fc2 where 4 D % 128 == 0; attn.qkv as a GEMM "
                                "of its own and mlp.fc1 where D % 192 == 0) fp16 hi*hi + two block-scaled fp8/fp6 corrections = 1.75 matrix units per 128 k "
                                "(matrix_units_per_product: issued units per algorithmic product, K padding included)",
                   "normalise": ("channel-sharded + all-gather of the planes (RIBCA_NORM_SHARD=1)" if (world > 1 and not args.impute and os.environ.get("RIBCA_NORM_SHARD") == "1")
                                 else "replicated on every rank" if world > 1 and not args.impute else "single rank"),
                   "parallelism": ("single GPU" if world == 1 else f"one tile per rank x {world} (replicas only)" if args.impute
                                   else f"cells sharded over {world} rank(s), one all-gather of per-cell probabilities")},
        "vit_gflop_per_cell": round(flops_cell / 1e9, 4),
        "vit_mfma_util_vs_bf16_dense": round(value * flops_cell / (world * PEAK_BF16_DENSE_TFLOPS * 1e12), 5),
        # the algorithmic fraction of the 16-bit dense peak cannot exceed 1 / (matrix units issued per product), FLOP-weighted over the products
        "mfma_cap_by_units": round(1.0 / avg_units({n: m.D for n, m in models.items()}, {n: m.depth for n, m in models.items()}), 4),
        "kernel_source_sha256": lib_sha256(),
    }
    # where a step's time goes, stage by stage (events on the stream the stage is enqueued on, averaged over the timed steps; a stage that
    # ends in a host read -- the label range, the percentile scalars, the final D2H -- includes the wait for it).  With cells sharded over
    # ranks the first two stages are REPLICATED work (every rank normalises the tile and builds the label table): the serial term of
    # the scaling curve.
    per_stage = {k: 0.0 for k in STAGES}
    for marks in stage_events:
        for name, a, b in marks:
            per_stage[name] += a.elapsed_time(b)
    out["per_stage_ms"] = {k: round(v / max(len(stage_events), 1), 3) for k, v in per_stage.items() if v > 0.0 or k in ("normalise", "label_table", "crop", "vit", "vote", "d2h")}
    out["replicated_preprocessing_ms"] = round((per_stage["normalise"] + per_stage["label_table"]) / max(len(stage_events), 1), 3)
    out["matrix_units_per_product"] = {name: matrix_units(m.D) for name, m in models.items()}
    out["cells_re_evaluated_at_full_precision"] = {name: m.last_recheck["cells"] for name, m in models.items()
                                                   if getattr(m, "last_recheck", None) and name not in vote_pair}
    # the vote's two classifiers, by the product's criterion (ops.decision_distance < VitModel.RECHECK_MARGIN on the pair's tables: top-2
    # margin over both tables, "Others", the confidence threshold); within_noise_floor = cells still within 2e-4 of a boundary after the
    # re-evaluation -- the cells whose label two correct fp32 evaluations need not agree on (Annotator.NOISE_FLOOR)
    out["cells_re_evaluated_at_full_precision"]["vote_pair (" + " + ".join(k for k in vote_pair if k) + ")"] = recheck_counts.get("vote_pair_cells_re_evaluated")
    out["cells_undecidable"] = recheck_counts.get("vote_pair_cells_within_noise_floor")
    out["recheck_margin"] = ops.VitModel.RECHECK_MARGIN
    # whether each classifier's weights were accepted for the MX products: the largest move of a logit difference between the fast and the
    # full-precision forward on a fixed 64-cell probe, measured at load time; predicted worst |dp| = 3 x delta / 4 against RECHECK_MARGIN / 2.5
    # (a model beyond the bar runs every product at three fp16 passes)
    out["mx_probe_logit_delta"] = {name: m.probe_logit_delta for name, m in models.items()}
    out["mx_probe_predicted_worst_dp"] = {name: m.probe_predicted_dp for name, m in models.items()}
    out["mx_fast_path_in_use"] = {name: m.uses_mx for name, m in models.items()}
    if imputer is not None:      # the imputer's own load-time probe: its imputed plane against the fp16x3 imputer's on 64 probe cells (bar 1e-3)
        out["imputer_probe_plane_delta"] = imputer.probe_plane_delta
        out["imputer_fast_path_in_use"] = imputer.fast_ok
    out["parity_audit"] = parity_audit_record(out["kernel_source_sha256"])
    if sharded:
        ag_ms = sum(a.elapsed_time(b) for a, b in ag_events)
        out["collective"] = collective_record(world, ag_ms, args.steps, ag_state.get("local", torch.zeros(0)))
        if out["collective"]["world_size_seen"] != args.gpus:
            # a line labelled n_gpus = N whose all-gather ran in a group of another size is not an N-GPU measurement: no line at all
            print(f"bench.py: the all-gather ran in a process group of {out['collective']['world_size_seen']} ranks, --gpus says {args.gpus}", file=sys.stderr)
            sys.exit(3)

    # ---- roofline of the dominant kernel (the bf16x3 GEMM family), one extra profiled pass -----------------------------
    if not args.no_roofline and rank == 0:
        # per-kernel durations: one stream, so a launch's events bracket that launch alone; one model at a time, so that every
        # model's GEMMs can be priced against the roofline that binds THEM (arithmetic intensity below / above the ridge)
        lo, hi = dist.shard_bounds(n_cells, rank, world) if sharded else (0, n_cells)
        n_local = hi - lo
        prof, per_model, shape_rows = {}, [], {}
        # the per-cell fused qkv + attention kernel (D <= 384) belongs to the family: it carries the qkv product of those classifiers
        GEMM_OPS = ("gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2", "cell_qkv_attention")
        fused_attn = fused_attn_on()
        for name, model in models.items():
            ops.prof_enable(True)
            one_pass(streams=1, models_sel=[name], gather=False)
            torch.cuda.synchronize()
            pm = ops.prof_read()
            ops.prof_enable(False)
            for k, v in pm.items():
                prof[k] = (prof.get(k, (0.0, 0))[0] + v[0], prof.get(k, (0, 0))[1] + v[1])
            d = model.D
            # per GEMM shape (classifiers of one width share their shapes): the launches of one op of one model, full blocks + the last block's
            # CLS-row launches, against their algorithmic FLOP -- the row the vendor yardstick below is put beside
            u_d = matrix_units(d)
            for op, kk, nn in (("qkv", d, 3 * d), ("proj", d, d), ("fc1", d, 4 * d), ("fc2", 4 * d, d)):
                if op == "qkv" and fused_attn and d <= 384:
                    continue        # runs inside the fused per-cell kernel: not a GEMM launch of its own
                fl = n_local * 2.0 * kk * nn * ((model.depth - 1) * 101 + (101 * 2.0 / 3.0 + 1.0 / 3.0 if op == "qkv" else 1.0))
                row = shape_rows.setdefault(f"{op}@{d}", {"M_per_launch": model.effective_chunk(args.chunk) * 101, "K": kk, "N": nn, "ms": 0.0, "flop": 0.0,
                                                          "matrix_units": round(u_d[op], 4)})
                row["ms"] += pm["gemm_" + op][0]
                row["flop"] += fl
            m_ms = sum(pm[k][0] for k in GEMM_OPS if k in pm)
            m_fl = n_local * ((model.depth - 1) * 24.0 * 101 * d * d + 6.0 * 101 * d * d + 18.0 * d * d)
            if fused_attn and d <= 384:      # its attention FLOPs run inside the family's kernel
                m_fl += n_local * (model.depth - 1) * 4.0 * 101 * 101 * d
            # A in + output out + z read / written (gemm_bytes_per_row: 4 B per element packed-split, 3 B in MX3)
            units = matrix_units(d)
            mu = (3.0 * units["qkv"] + units["proj"] + 4.0 * units["fc1"] + 4.0 * units["fc2"]) / 12.0
            m_by = n_local * 101 * (model.depth - 1) * gemm_bytes_per_row(d)
            ai = m_fl / m_by
            # ridge of the issued work: this classifier's matrix units per product against the 16-bit dense peak, HBM at 8 TB/s
            ridge = PEAK_BF16_DENSE_TFLOPS * 1e12 / mu / 8.0e12
            tf, gbs = m_fl / (m_ms * 1e-3) / 1e12, m_by / (m_ms * 1e-3) / 1e9
            bound = "mfma" if ai >= ridge else "hbm"
            per_model.append({"model": name, "D": d, "gemm_ms": round(m_ms, 2), "algorithmic_tflops": round(tf, 1),
                              "algorithmic_gb_per_s": round(gbs, 1), "flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(ridge, 1),
                              "bound": bound, "matrix_units": round(mu, 3),
                              "frac_of_bound": round(mu * tf / PEAK_BF16_DENSE_TFLOPS if bound == "mfma" else gbs / 8000.0, 4)})
        # ---- vendor yardstick, measured in THIS run on THIS box, outside the product path: torch.matmul (hipBLASLt) in plain
        # fp16 on the same [M x K] . [K x N] shapes at the same M per launch.  A plain 16-bit GEMM is ONE matrix unit per product: the figure to
        # hold against it is a product kernel's ISSUED rate (algorithmic TFLOP/s x its matrix units per product).
        per_shape = {}
        if args.impute:
            shape_rows = {}      # (the imputer's GEMMs are timed under the same classes as immune_full's: no per-shape rows for this workload)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import vendor_gemm
            by_cells = {}
            for key, row in shape_rows.items():
                if key in {n_ for n_, _, _ in vendor_gemm.SHAPES}:
                    by_cells.setdefault(row["M_per_launch"] // 101, []).append(key)
            vendor = {}
            for cells_v, keys in by_cells.items():
                vendor.update(vendor_gemm.measure(cells_v, names=keys, dtypes=("fp16",), rounds=2)["fp16"])
        except Exception as e:      # the yardstick must never cost the line
            vendor = {}
            print(f"bench.py: vendor yardstick skipped ({type(e).__name__}: {e})", file=sys.stderr)
        for key, row in sorted(shape_rows.items(), key=lambda kv: -kv[1]["ms"]):
            alg = row["flop"] / (row["ms"] * 1e-3) / 1e12 if row["ms"] > 0 else 0.0
            per_shape[key] = {"M_per_launch": row["M_per_launch"], "K": row["K"], "N": row["N"], "ms_per_pass": round(row["ms"], 2),
                              "algorithmic_tflops": round(alg, 1), "matrix_units_per_product": row["matrix_units"],
                              "issued_tflops": round(alg * row["matrix_units"], 1),
                              "vendor_fp16_gemm_tflops": vendor.get(key, {}).get("tflops"),
                              "issued_over_vendor": round(alg * row["matrix_units"] / vendor[key]["tflops"], 3) if key in vendor else None}
        # one number for the comparison: the shapes the yardstick covers, at the product's time and at the time the SAME matrix units would take
        # at the vendor's plain-GEMM rate on each shape (no epilogue work counted for the vendor)
        cov = [v for v in per_shape.values() if v["vendor_fp16_gemm_tflops"]]
        vendor_summary = None
        if cov:
            ours_ms = sum(v["ms_per_pass"] for v in cov)
            at_vendor_ms = sum(v["ms_per_pass"] * v["issued_over_vendor"] for v in cov)
            vendor_summary = {"shapes": len(cov), "product_ms_per_pass": round(ours_ms, 1), "same_matrix_units_at_vendor_rate_ms_per_pass": round(at_vendor_ms, 1),
                              "product_speed_vs_vendor_rate": round(at_vendor_ms / ours_ms, 3)}
        gemm_flops = 0.0
        for name, model in models.items():
            d = model.D
            # qkv + proj + fc1 + fc2 (algorithmic, unpadded); in the last block proj / fc1 / fc2 run on the CLS row only
            gemm_flops += n_local * ((model.depth - 1) * 24.0 * 101 * d * d + 6.0 * 101 * d * d + 18.0 * d * d)
            if fused_attn and d <= 384:
                gemm_flops += n_local * (model.depth - 1) * 4.0 * 101 * 101 * d
        g_ms = sum(prof[k][0] for k in GEMM_OPS if k in prof)
        g_n = sum(prof[k][1] for k in GEMM_OPS if k in prof)
        achieved = gemm_flops / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        # committed rocprofv3 evidence for the same kernels (tools/collect_profiles.sh, tools/collect_pmc_sq.sh): HBM/fabric bytes per
        # launch from the FETCH_SIZE / WRITE_SIZE passes, MFMA-busy and LDS-active fractions from the SQ counter passes.  Both files
        # carry the fingerprint of the kernel sources they were measured on: other sources being timed here -> null, not stale numbers
        traffic, traffic_src, busy, lds, pass_bytes = None, None, None, None, None
        sha = out["kernel_source_sha256"]
        tpath = os.path.join(ROOT, "profiles", PROFILE_ROUND, "gemm_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("kernel_source_sha256") == sha:
                traffic = round(tj["traffic_bytes_per_launch"])
                traffic_src = f"profiles/{PROFILE_ROUND}/gemm_traffic.json"
                if tj.get("vit_bytes_per_cell"):
                    pass_bytes = tj["vit_bytes_per_cell"] * n_local
        spath = os.path.join(ROOT, "profiles", PROFILE_ROUND, "sq_summary.json")
        if os.path.exists(spath):
            sq = json.load(open(spath))
            if sq.get("kernel_source_sha256") == sha:
                gem = [v for k, v in sq.items() if k.startswith(("gemm_ps_split_kernel", "gemm_ps_duo_kernel", "gemm_mx_duo_kernel", "cell_qkv_attention_kernel"))]
                cyc = sum(v["kernel_cycles"] for v in gem)
                if cyc > 0:
                    busy = round(sum(v["mfma_busy_frac"] * v["kernel_cycles"] for v in gem) / cyc, 4)
                    lds = round(sum(v["lds_active_frac"] * v["kernel_cycles"] for v in gem) / cyc, 4)
        alg_bytes = 0.0
        for name, model in models.items():      # algorithmic bytes per pass of the four GEMMs: A read once, output written once, z RMW
            d = model.D
            alg_bytes += n_local * 101 * (model.depth - 1) * gemm_bytes_per_row(d) + 12.0 * d * d * 4.0 * model.depth * ((n_local + args.chunk - 1) // args.chunk)
        # the bound that binds most of the GEMM time: every classifier's GEMMs are priced against their own roofline (per_model_*),
        # the family's label is the time-weighted majority
        mu_all = avg_units({n: m.D for n, m in models.items()}, {n: m.depth for n, m in models.items()})
        t_mfma = sum(m["gemm_ms"] for m in per_model if m["bound"] == "mfma")
        t_hbm = sum(m["gemm_ms"] for m in per_model if m["bound"] == "hbm")
        out["per_kernel_ms"] = {k: round(v[0], 3) for k, v in prof.items() if v[1]}
        out["per_model_gemm_ms"] = {m["model"]: m["gemm_ms"] for m in per_model}
        out["per_model_bound"] = {m["model"]: m["bound"] for m in per_model}
        # ISSUED fraction (algorithmic TFLOP/s x matrix units per product / 2.5 PF) where MFMA-bound, algorithmic GB/s of 8 TB/s where HBM-bound:
        # not the algorithmic fraction of the peak, which is roofline.frac
        out["per_model_issued_frac_of_bound"] = {m["model"]: m["frac_of_bound"] for m in per_model}
        out["per_model_flop_per_byte"] = {m["model"]: m["flop_per_byte"] for m in per_model}
        # whole ViT pass against HBM: counter bytes (GEMM + attention + statistics kernels, 2 x FETCH_SIZE + WRITE_SIZE of the
        # committed, sha-matched passes, scaled per cell) over the timed step
        out["hbm_gb_per_s_pass"] = round(pass_bytes / (ms_per_step * 1e-3) / 1e9, 1) if pass_bytes else None
        out["hbm_frac_of_8tbs_pass"] = round(pass_bytes / (ms_per_step * 1e-3) / 8.0e12, 4) if pass_bytes else None
        out["roofline"] = {"bound": "mfma" if t_mfma >= t_hbm else "hbm",
                           "bound_note": f"time-weighted over the classifiers: {t_mfma:.0f} ms of GEMMs MFMA-bound ({', '.join(m['model'] for m in per_model if m['bound'] == 'mfma')}), "
                                         f"{t_hbm:.0f} ms HBM-bound ({', '.join(m['model'] for m in per_model if m['bound'] == 'hbm')}); "
                                         "a classifier's side of the ridge follows its algorithmic FLOP per byte (per_model_flop_per_byte) against 2.5 PF / its matrix units per product / 8 TB/s",
                           "kernel": "gemm_mx_duo_kernel (fp16 hi*hi + block-scaled fp8/fp6 corrections, 3-byte activations: mlp.fc2 at D = 288 / 384 / 576, "
                                     "mlp.fc1 at D = 384 / 576, attn.qkv at D = 576) + gemm_ps_duo_kernel (fp16x3: attn.proj with the residual tile through the "
                                     "operand ring -- writing the new rows in MX3 as well at D = 384 / 576 --, fc1 at D = 144 / 288, fc2 at D = 144) + "
                                     "gemm_ps_split_kernel (the last blocks' CLS rows) + cell_qkv_attention_kernel (norm1 + qkv + attention, D <= 384, fp16x3)",
                           "achieved": round(achieved, 2),
                           "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 5),
                           "traffic": traffic, "traffic_unit": f"bytes/launch (2*FETCH_SIZE + WRITE_SIZE, {traffic_src})",
                           # provenance of the counter-derived fields (traffic, mfma_busy_frac, lds_active_frac, hbm_*_pass): rocprofv3 --pmc passes
                           # cannot run inside this process, so they are READ from the committed profile of the same kernel sources (sha-matched:
                           # other sources -> null); achieved / frac / avg_launch_ms / per_kernel_ms are measured live by this run's own events
                           "counter_fields_source": (f"committed profile (profiles/{PROFILE_ROUND}/, kernel_source_sha256 matches)" if traffic is not None or busy is not None
                                                     else "none: no committed counter passes for these kernel sources"),
                           "timing_fields_source": "measured in this run (HIP events on the launch stream)",
                           "algorithmic_bytes_per_launch": round(alg_bytes / max(g_n, 1)),
                           # per GEMM shape: this run's launches beside the vendor library's plain fp16 GEMM of the same shape, timed in this run
                           "per_shape": per_shape,
                           "vendor_rate_summary": vendor_summary,
                           "per_shape_note": "issued_tflops = algorithmic_tflops x matrix_units_per_product (the work the matrix cores are handed); "
                                             "vendor_fp16_gemm_tflops = torch.matmul (hipBLASLt) in plain fp16, ONE unit per product, same M / K / N, "
                                             "operands rotated, measured in this run outside the product path (tools/vendor_gemm.py); the product kernels also "
                                             "carry the LayerNorm fold, GELU / residual / statistics epilogues and the MX3 emission the vendor GEMM does not",
                           "mfma_busy_frac": busy, "lds_active_frac": lds,
                           "counters": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), SQ_LDS_IDX_ACTIVE / (256 CUs x ...): "
                                       f"profiles/{PROFILE_ROUND}/sq_summary.json; null = the committed counters belong to another build of the library",
                           "launches": int(g_n), "avg_launch_ms": round(g_ms / max(g_n, 1), 5),
                           "algorithmic_gflop_per_launch": round(gemm_flops / max(g_n, 1) / 1e9, 4),
                           "matrix_units_per_product": round(mu_all, 4),      # FLOP-weighted over the classifiers (per shape: top-level matrix_units_per_product)
                           "issued_mfma_frac_of_peak": round(mu_all * achieved / PEAK_BF16_DENSE_TFLOPS, 4),
                           "per_model_note": "GEMMs of one classifier: algorithmic FLOP per algorithmic byte against the ridge of the ISSUED "
                                             "work (the classifier's matrix units per product at 2.5 PF dense / 8 TB/s: 104 FLOP/B at 3 units); "
                                             "per_model_issued_frac_of_bound = ISSUED MFMA fraction of peak where MFMA-bound, algorithmic GB/s of 8 TB/s where "
                                             "HBM-bound (top-level per_model_* keys)"}

    # ---- the boundary itself: Annotator.preprocess -> predict -> export_annotations from host files ------------------------
    if not args.no_dropin and rank == 0 and world == 1 and not args.impute:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # the Annotator prints the reference's panel messages: keep stdout to ONE JSON line
            out["dropin"] = dropin_bench(args, raw, mask, markers, models, srcs, one_pass_models=lambda sel: one_pass(models_sel=sel))

    # ---- BASELINE config 5 through the product class: a batch CSV of one tile per rank, Annotator in tile-per-rank mode (every rank) -------
    if not args.no_dropin and args.impute:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):
            rec = dropin_tiles_bench(args, raw, mask, markers, models, imputer, rank, world, backend, dev)
        if rank == 0:
            out["dropin"] = rec

    # ---- CPU baseline: the oracle (numpy/scipy/torch fp32 restatement of the reference path) on a bounded sample ------------
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, raw, mask, markers, seed, n_cells)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        # the other ranks wait here while rank 0 runs its profiled passes: every rank then tears the group down together (a communicator
        # destroyed on some ranks while another still holds it is the kind of exit RCCL may not take quietly)
        tdist.barrier()
        tdist.destroy_process_group()


def fused_attn_on():
    """RIBCA_CELL_ATTN as the library reads it (atoi)"""
    try:
        return int(os.environ.get("RIBCA_CELL_ATTN", "1")) != 0
    except ValueError:
        return False


def matrix_units(d):
    """matrix units (one unit = one pass of the 16-bit dense rate) ISSUED per algorithmic product of the four Linears of a full block at
    width d: three fp16 passes, or 1.75 (per 128 k: 4 x f16 = one unit, fp8 x fp6 = half a unit, fp6 x fp6 = a quarter) on the MX kernel -- times
    the K padding where the MX kernel pads D to a multiple of 128 (qkv / fc1 at D = 576: 640 / 576).  qkv inside the fused per-cell
    kernel (D <= 384) stays at three passes."""
    from multiplexed_image_annotator_amd import _lib
    mx = bool(_lib.lib().ribca_mx_enabled(int(d)))
    mxz = bool(_lib.lib().ribca_mxz_enabled(int(d)))
    pad = ((d + 127) // 128 * 128) / float(d)
    qkv_own = not (fused_attn_on() and d <= 384)
    return {"qkv": 1.75 * pad if (mxz and qkv_own) else 3.0, "proj": 3.0, "fc1": 1.75 * pad if mxz else 3.0, "fc2": 1.75 if mx else 3.0}


def gemm_bytes_per_row(d):
    """algorithmic bytes per token row and full block of the four Linears' launches: A read once, output written once, z read and written
    by proj / fc2 (4 B per element packed-split, 3 B in MX3: fp16 hi + e4m3 lo + scale bytes)"""
    from multiplexed_image_annotator_amd import _lib
    mx = bool(_lib.lib().ribca_mx_enabled(int(d)))
    mxz = bool(_lib.lib().ribca_mxz_enabled(int(d)))
    qkv_own = not (fused_attn_on() and d <= 384)
    hb = 3.0 if mx else 4.0                                        # bytes per element of h
    zb = 3.0 if mxz else 4.0                                       # bytes per element of the residual rows qkv (own GEMM) / fc1 read
    qkv = ((zb if qkv_own else 4.0) + (3 * 4.0 if qkv_own else 4.0)) * d      # fused with attention: z in, attention output out
    proj = 4.0 * d * (1 + 2) + (3.0 * d if mxz else 0.0)            # + the MX3 copy of the new rows
    fc1 = zb * d + hb * 4 * d
    fc2 = hb * 4 * d + 4.0 * d * 2 + (3.0 * d if (mxz and qkv_own) else 0.0)
    return qkv + proj + fc1 + fc2


def avg_units(dims, depths):
    """FLOP-weighted matrix units per product over the classifiers' full blocks (qkv 3 D^2, proj D^2, fc1 4 D^2, fc2 4 D^2 per row)"""
    num = den = 0.0
    for name, d in dims.items():
        u = matrix_units(d)
        w = {"qkv": 3.0, "proj": 1.0, "fc1": 4.0, "fc2": 4.0}
        blocks = max(depths[name] - 1, 1)
        num += blocks * d * d * sum(w[k] * u[k] for k in w)
        den += blocks * d * d * sum(w.values())
    return num / den


PROFILE_ROUND = "r6"      # the directory under profiles/ whose counter passes the line may quote (sha-matched)


def parity_audit_record(sha):
    """label flips / undecidable cells of the committed config-3 parity audit (tests/test_gpu_e2e.py::test_config3_parity_audit_2000_cells
    writes it on the GPU box) -- quoted only when it was taken on these kernel sources"""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, "parity_audit_config3.json")
    try:
        j = json.load(open(path))
    except (OSError, ValueError):
        return None
    if j.get("kernel_source_sha256") != sha:
        return {"source": f"profiles/{PROFILE_ROUND}/parity_audit_config3.json belongs to other kernel sources", "label_flips_in_audit": None}
    return {"source": f"committed profile (profiles/{PROFILE_ROUND}/parity_audit_config3.json, kernel_source_sha256 matches)",
            "cells_audited_per_model": j.get("cells_per_model"), "label_flips_in_audit": j.get("label_flips_total"),
            "flips_outside_twice_the_error_band": j.get("flips_outside_band_total"), "max_abs_dp": j.get("max_abs_dp")}


def lib_sha256():
    """fingerprint of the kernel sources + flags the timed library is built from (build.source_fingerprint; __graft_entry__.build()
    has just rebuilt the library if any of them changed).  The committed counter files carry the one they were measured on."""
    from multiplexed_image_annotator_amd import build as _build
    try:
        return _build.source_fingerprint()
    except OSError:
        return None


def config1_bench(args):
    """BASELINE.json configs[0] (reference main.py:9-36 on examples/example_1.*, batch_size=8, device=cpu) through its stand-in
    (the reference's example_1 mask, 1850 cells, + the seeded 7-channel Basic-panel image; the .tif is a missing
    blob).  GPU: the drop-in Annotator end to end from host .npy files; CPU: the oracle pipeline timed in full on the same inputs."""
    import shutil
    import tempfile
    import __graft_entry__
    __graft_entry__.build()
    from multiplexed_image_annotator_amd import _lib, synth
    from multiplexed_image_annotator_amd.annotator import Annotator
    _lib.require_gpu()
    gdir = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gdir, "config1.json")))
    arrs = np.load(os.path.join(gdir, "config1.npz"))
    mask = np.load(os.path.join(gdir, "cellpos.npz"))["example1_mask"].astype(np.int32)
    raw = synth.make_image_for_mask(torch.from_numpy(mask), len(meta["markers"]), meta["seed"]).numpy().astype(np.uint16)
    sd = synth.make_vit_state_dict("immune_base", meta["seed"])
    sd["head.bias"] = torch.from_numpy(arrs["head_bias"])
    tmp = tempfile.mkdtemp(prefix="ribca_c1_")
    try:
        np.save(os.path.join(tmp, "img.npy"), raw)
        np.save(os.path.join(tmp, "mask.npy"), mask)
        mf = os.path.join(tmp, "markers.txt")
        with open(mf, "w") as f:
            f.write("\n".join(meta["markers"]) + "\n")
        with open(os.path.join(tmp, "images.csv"), "w") as f:
            f.write("image_path,mask_path\n%s,%s\n" % (os.path.join(tmp, "img.npy"), os.path.join(tmp, "mask.npy")))
        shared = {}

        def run_once():
            a = Annotator(mf, os.path.join(tmp, "images.csv"), "cuda", tmp, "c1", True, False, -1, True, meta["blur"], meta["amax"], meta["conf"],
                          30, None)
            if shared:
                a.models, a._loaded = dict(shared), True
            else:
                a.set_weights({"immune_base": sd})
            a.preprocess()
            a.predict(meta["batch_size"])
            a.export_annotations()
            shared.update(a.models)
            labels = list(a.annotations[0])
            a.clear_tmp()
            a.logger.close()
            return labels

        for _ in range(max(args.warmup, 1)):
            labels = run_once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            labels = run_once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        n = len(labels)
        out = {"metric": "cells/sec annotated (whole node)", "value": round(n / dt, 2), "unit": "cells/s", "n_gpus": 1, "steps": args.steps,
               "warmup": max(args.warmup, 1), "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f16", "data": "synthetic image on the reference's examples/example_1_cell_mask.png",
               "config": {"workload": "BASELINE configs[0] stand-in: 600x600 example_1 mask, 1850 cells, 7-marker Basic panel, immune_base ViT, "
                                      "Annotator.preprocess -> predict(8) -> export_annotations from host .npy files",
                          "baseline_config": "configs[0]", "cells": n},
               "labels_identical_to_reference_golden": labels == meta["labels"]}
        if not args.no_cpu_baseline:
            from oracle import ref_pipeline
            threads = min(16, len(os.sched_getaffinity(0)))
            torch.set_num_threads(threads)
            t0 = time.perf_counter()
            r = ref_pipeline.run_image(raw, mask, mf, {"immune_base": sd}, strict=True, normalize=True, blur=meta["blur"], amax=meta["amax"],
                                       confidence=meta["conf"], batch_size=meta["batch_size"])
            t_cpu = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(n / t_cpu, 3), "unit": "cells/s", "cores": threads, "kind": "port",
                                   "sample": f"the whole config (all {n} cells, batch_size 8) timed in full: {t_cpu:.1f} s; labels equal the "
                                             f"reference golden: {r['labels'] == meta['labels']}"}
        print(json.dumps(out))
        return 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def dropin_bench(args, raw_dev, mask_dev, markers, models, srcs, one_pass_models):
    """Times the drop-in class itself on the same tile: inputs are HOST files (uint16 .npy image, int32 .npy mask, marker list),
    the timed region is ``Annotator(...)`` -> ``preprocess()`` (file read, H2D, normalise, label table, crop + intensity table)
    -> ``predict()`` -> ``export_annotations()`` (CSV on disk) -> ``clear_tmp()``.  With this 15-marker file the reference's model
    choice (model.py:241-349) runs ONE classifier, immune_full (struct / nerve markers cannot be named by a 15-marker panel), so
    the same model set is also timed through the kernel-path ``one_pass`` for a like-for-like ratio.  Packed weights stay
    resident across passes, as in the kernel path."""
    import shutil
    import tempfile
    from multiplexed_image_annotator_amd.annotator import Annotator
    tmp = tempfile.mkdtemp(prefix="ribca_dropin_")
    try:
        np.save(os.path.join(tmp, "img.npy"), raw_dev.cpu().numpy().view(np.uint16))
        np.save(os.path.join(tmp, "mask.npy"), mask_dev.cpu().numpy())
        with open(os.path.join(tmp, "markers.txt"), "w") as f:
            f.write("\n".join(markers) + "\n")
        with open(os.path.join(tmp, "images.csv"), "w") as f:
            f.write("image_path,mask_path\n%s,%s\n" % (os.path.join(tmp, "img.npy"), os.path.join(tmp, "mask.npy")))

        def run_once():
            a = Annotator(os.path.join(tmp, "markers.txt"), os.path.join(tmp, "images.csv"), "cuda", tmp, "bench", True, False, -1, True, 0.3,
                          99.8, 0.3, 30, None)
            a.chunk_cells, a.streams = args.chunk, args.streams
            a.models = {k: v for k, v in models.items() if k == "immune_full"}
            a._loaded = True
            a.preprocess()
            a.predict(128)
            a.export_annotations()
            n = len(a.annotations[0])
            a.clear_tmp()
            a.logger.close()
            return n

        run_once()
        torch.cuda.synchronize()
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            n = run_once()
        torch.cuda.synchronize()
        t_api = (time.perf_counter() - t0) / reps
        one_pass_models(["immune_full"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            one_pass_models(["immune_full"])
        torch.cuda.synchronize()
        t_ops = (time.perf_counter() - t0) / reps
        return {"value": round(n / t_api, 2), "unit": "cells/s", "ms_per_tile": round(t_api * 1e3, 2), "models": ["immune_full"],
                "includes": "np.load of image + mask, H2D, normalise, label table, crop + intensity table, ViT, vote, CSV write",
                "kernel_path_same_models": round(n / t_ops, 2), "ratio_to_kernel_path": round(t_ops / t_api, 4)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def dropin_tiles_bench(args, raw_dev, mask_dev, markers, models, imputer, rank, world, backend, dev):
    """BASELINE config 5 through the drop-in class (reference main.py:39-52 batch_run): ONE batch CSV listing every rank's tile, every rank
    an ``Annotator`` in tile-per-rank mode (dist.tile_mode: at least one image per rank -> whole images per rank, replicas only, nothing
    exchanged on the data path; a single rank simply annotates its one tile), infer=True with one full-panel marker missing -> MAE
    imputer + immune_full.  Timed per rank: Annotator(...) -> preprocess() (file read, H2D, normalise, label table, crop) -> predict()
    (imputer, ViT, re-evaluation, vote) -> export_annotations() (this rank's CSV); value = all ranks' cells / the slowest rank's time."""
    import shutil
    import tempfile
    import torch.distributed as tdist
    from multiplexed_image_annotator_amd.annotator import Annotator
    box = [tempfile.mkdtemp(prefix="ribca_tiles_") if rank == 0 else None]
    if world > 1:
        tdist.broadcast_object_list(box, src=0)
    tmp = box[0]
    try:
        np.save(os.path.join(tmp, f"img{rank}.npy"), raw_dev.cpu().numpy().view(np.uint16))
        np.save(os.path.join(tmp, f"mask{rank}.npy"), mask_dev.cpu().numpy())
        if rank == 0:
            with open(os.path.join(tmp, "markers.txt"), "w") as f:
                f.write("\n".join(markers) + "\n")
            with open(os.path.join(tmp, "images.csv"), "w") as f:
                f.write("image_path,mask_path\n" + "".join("%s,%s\n" % (os.path.join(tmp, f"img{r}.npy"), os.path.join(tmp, f"mask{r}.npy")) for r in range(world)))
        if world > 1:
            tdist.barrier()

        def run_once():
            a = Annotator(os.path.join(tmp, "markers.txt"), os.path.join(tmp, "images.csv"), "cuda", tmp, "bench", False, True, -1, True, 0.3,
                          99.8, 0.3, 30, None)
            assert a.tile_mode == (world > 1)
            a.chunk_cells, a.streams = args.chunk, args.streams
            a.models = {k: v for k, v in models.items() if k == "immune_full"}
            a.imputers = {"immune_full": imputer}
            a._loaded = True
            a.preprocess()
            a.predict(128)
            a.export_annotations()
            n = sum(len(x) for x in a.annotations)
            a.clear_tmp()
            a.logger.close()
            return n

        run_once()
        torch.cuda.synchronize()
        if world > 1:
            tdist.barrier()
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            n = run_once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        total = n
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            dt = float(t.item())
            c = torch.tensor([n], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            tdist.all_reduce(c, op=tdist.ReduceOp.SUM)
            total = int(c.item())
            tdist.barrier()
        return {"value": round(total / dt, 2), "unit": "cells/s", "ms_per_tile": round(dt * 1e3, 2), "models": ["immune_full"], "imputer": True,
                "mode": f"tile-per-rank x {world} (Annotator.tile_mode, replicas only)" if world > 1 else "single rank, one tile",
                "includes": "np.load of image + mask, H2D, normalise, label table, crop + intensity table, MAE imputer, ViT, re-evaluation, vote, CSV write"}
    finally:
        if world > 1:
            tdist.barrier()
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)


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
    k = min(args.cpu_sample, len(ids))
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
        ref_vit.predict_proba(sd, x[:, :c], 128)
        t_vit += (time.perf_counter() - t) / k
        n_models += 1
    per_cell = (t_norm + t_label) / n_cells + n_models * t_crop + t_vit     # the reference crops once per applicable panel
    return {"value": round(1.0 / per_cell, 3), "unit": "cells/s", "cores": threads, "kind": "port",
            "sample": f"oracle on a {side}x{side} corner: normalise+label scan scaled by area to the full tile ({t_norm:.1f}s+{t_label:.2f}s), "
                      f"crop/soft-mask {t_crop*1e3:.2f} ms/cell/panel (1 thread, as the reference) and {n_models} fp32 ViTs "
                      f"{t_vit*1e3:.1f} ms/cell ({threads} threads, batch 128) on {k} cells"}


if __name__ == "__main__":
    main()
