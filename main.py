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


def run(args):
    import __graft_entry__
    __graft_entry__.build()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    from multiplexed_image_annotator_amd.annotator import Annotator
    csv_path = args.batch_csv
    if args.image_path:
        csv_path = os.path.join(args.main_dir, "images.csv")
        if int(os.environ.get("RANK", "0")) == 0:
            os.makedirs(args.main_dir, exist_ok=True)
            with open(csv_path, "w") as f:
                f.write("image_path,mask_path\n%s,%s\n" % (args.image_path, args.mask_path))
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            import torch.distributed as dist
            dist.barrier()
    a = Annotator(args.marker_list_path, csv_path, args.device, args.main_dir, args.batch_id, args.strict, args.infer, args.min_cells,
                  args.normalize, args.blur, args.amax, args.confidence, args.cell_size, args.cell_type_confidence, n_jobs=args.n_jobs)
    p = a.channel_parser
    if not (p.immune_base or p.immune_extended or p.immune_full or p.struct or p.nerve):
        raise ValueError("No panels are applied. Please check the marker list.")
    a.preprocess()
    a.predict(args.bs)
    a.export_annotations()
    n_cells = min((len(ids) for ids in a.preprocessor.cell_ids), default=0)
    if args.n_regions > 0 and n_cells >= 201:      # the reference's 201-neighbour query raises on smaller images
        a.tissue_region_analysis(args.n_regions)
        a.export_annotations()                      # reference order is regions -> export; the CSV gains its Tissue Region column
    if n_cells >= 25:                              # the reference's kNN (25 neighbours) raises on smaller images
        a.neighborhood_analysis(integrate=True, normalize=True)
    a.colorize(from_script=True)
    a.clear_tmp()
    return a


if __name__ == "__main__":
    run(parse_args())
