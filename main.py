#!/usr/bin/env python3
"""CLI with the reference's argument surface (reference main.py:56-112) on the MI355X hot path.

Runs marker parsing -> preprocess -> predict -> export_annotations -> tissue_region_analysis -> neighborhood_analysis ->
colorize, as the reference does (main.py:19-28); its plotting steps (heat-maps, pie charts, legends) are CPU work downstream
of the CSV and are not part of this accelerated path.  Multi-GPU: launch under ``python -m torch.distributed.run --nproc-per-node N main.py ...``.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description='Annotate cell types in multiplexed images (MI355X hot path)')
    ap.add_argument('--marker-list-path', type=str, required=True)
    ap.add_argument('--device', type=str, default='cuda')
    ap.add_argument('--main-dir', type=str, default='./')
    ap.add_argument('--batch-id', type=str, required=True)
    ap.add_argument('--strict', action='store_true')
    ap.add_argument('--infer', action='store_true', default=True)      # as in the reference: cannot be switched off here
    ap.add_argument('--no-infer', dest='infer', action='store_false', help='blank planes instead of marker imputation')
    ap.add_argument('--min-cells', type=int, default=-1)
    ap.add_argument('--n-regions', type=int, default=3)
    ap.add_argument('--normalize', action='store_true', default=True)
    ap.add_argument('--no-normalize', dest='normalize', action='store_false')
    ap.add_argument('--blur', type=float, default=0.3)
    ap.add_argument('--amax', type=float, default=99.8)
    ap.add_argument('--confidence', type=float, default=0.3)
    ap.add_argument('--cell-type-confidence', type=float, default=None)
    ap.add_argument('--bs', type=int, default=128)
    ap.add_argument('--cell-size', type=int, default=30)
    ap.add_argument('--n_jobs', type=int, default=0)
    grp = ap.add_mutually_exclusive_group(required=True)
    grp.add_argument('--image-path', type=str)
    grp.add_argument('--batch-csv', type=str)
    ap.add_argument('--mask-path', type=str)
    args = ap.parse_args(argv)
    if args.image_path and not args.mask_path:
        ap.error("--mask-path is required when using --image-path")
    return args


def _setup():
    import __graft_entry__
    __graft_entry__.build()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group(os.environ.get("RIBCA_DIST_BACKEND", "nccl"))


def _pipeline(annotator, bs, n_regions):
    """The call sequence of reference main.py:19-28 / 43-52: the CSV is exported ONCE, before the tissue-region analysis, so its
    "Tissue Region" column reads None exactly as the reference's does (RIBCA_EXPORT_REGIONS=1 opts in to a second export that
    carries the regions).  Two guards the reference lacks keep small images from raising inside its k-NN queries.  Its plotting
    calls (generate_heatmap, cell_type_composition) are outside the accelerated path: this Annotator logs and skips them."""
    p = annotator.channel_parser
    if not p.immune_base and not p.immune_extended and not p.immune_full and not p.struct and not p.nerve:
        raise ValueError("No panels are applied. Please check the marker list.")
    annotator.preprocess()
    annotator.predict(bs)
    annotator.generate_heatmap(integrate=True)
    annotator.export_annotations()
    n_cells = annotator.min_cells_per_image()
    if n_regions > 0 and n_cells >= 201:           # the reference's 201-neighbour query raises on smaller images
        annotator.tissue_region_analysis(n_regions)
        if os.environ.get("RIBCA_EXPORT_REGIONS") == "1":      # opt-in deviation: the reference's CSV never carries the regions
            annotator.export_annotations()
    if n_cells >= 25:                               # the reference's kNN (25 neighbours) raises on smaller images
        annotator.neighborhood_analysis(integrate=True, normalize=True)
    annotator.colorize(from_script=True)
    annotator.cell_type_composition()
    annotator.clear_tmp()


def run(marker_list_path, image_path, mask_path, device, main_dir, batch_id, bs, strict, infer, min_cells, n_regions, normalize, blur, amax,
        confidence, cell_size, cell_type_confidence, n_jobs):
    """reference main.py:9-36: one image + mask -> images.csv -> annotate; returns (intensity_dict, names) as the reference does."""
    import numpy as np
    _setup()
    from multiplexed_image_annotator_amd.annotator import Annotator
    path_ = os.path.join(main_dir, "images.csv")
    if int(os.environ.get("RANK", "0")) == 0:
        os.makedirs(main_dir, exist_ok=True)
        with open(path_, "w") as f:
            f.write("image_path,mask_path\n%s,%s\n" % (image_path, mask_path))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        dist.barrier()
    annotator = Annotator(marker_list_path, path_, device, main_dir, batch_id, strict, infer, min_cells, normalize, blur, amax, confidence,
                          cell_size, cell_type_confidence, n_jobs=n_jobs)
    _pipeline(annotator, bs, n_regions)
    intensity_dict = {}
    full = annotator.preprocessor.intensity_full[0]
    for i in range(len(full)):
        intensity_dict[i + 1] = full[i]
    intensity_dict[0] = np.zeros_like(full[0])
    names = annotator.get_cell_type_names()
    return intensity_dict, names


def batch_run(marker_list_path, image_path, device, main_dir, batch_id, bs, strict, infer, min_cells, n_regions, normalize, blur, amax,
              confidence, cell_size, cell_type_confidence, n_jobs=0):
    """reference main.py:39-52: ``image_path`` is a CSV with columns image_path,mask_path."""
    _setup()
    from multiplexed_image_annotator_amd.annotator import Annotator
    annotator = Annotator(marker_list_path, image_path, device, main_dir, batch_id, strict, infer, min_cells, normalize, blur, amax,
                          confidence, cell_size, cell_type_confidence, n_jobs=n_jobs)
    _pipeline(annotator, bs, n_regions)


def main(argv=None):
    args = parse_args(argv)
    common = dict(marker_list_path=args.marker_list_path, device=args.device, main_dir=args.main_dir, batch_id=args.batch_id, bs=args.bs,
                  strict=args.strict, infer=args.infer, min_cells=args.min_cells, n_regions=args.n_regions, normalize=args.normalize,
                  blur=args.blur, amax=args.amax, confidence=args.confidence, cell_size=args.cell_size,
                  cell_type_confidence=args.cell_type_confidence, n_jobs=args.n_jobs)
    if args.batch_csv:
        return batch_run(image_path=args.batch_csv, **common)
    return run(image_path=args.image_path, mask_path=args.mask_path, **common)


if __name__ == "__main__":
    main()
