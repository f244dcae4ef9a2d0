"""``Annotator``: drop-in for the reference orchestrator on its hot path (cell_type_annotation/model.py:90-919):
same constructor, ``preprocess()``, ``predict(batch_size)``, ``export_annotations()``, ``clear_tmp()``,
``get_cell_type_names()`` and the attributes downstream code reads (``annotations``, ``confidence``, ``annotations_all``,
``cell_types``, ``channel_parser``, ``preprocessor``, ``*_pred``).  Compute runs in the HIP library; the plotting methods of
the reference (heat-maps, pie charts, UMAP) are outside the accelerated path: they log and return, so the
reference's own call sequence (main.py:19-28) completes against this class.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib, colors, dist, ops
from .logger import Logger
from .marker_parse import MarkerParser
from .preprocess import ImageProcessor

#: per-model class order (model.py:247-252, 266-270, 284-287, 309-312, 334)
CLASS_NAMES: Dict[str, List[str]] = {
    "immune_full": ["CD4 T cell", "CD8 T cell", "Dendritic cell", "B cell", "M1 macrophage cell", "M2 macrophage cell",
                    "Regulatory T cell", "Granulocyte cell", "Plasma cell", "Natural killer cell", "Mast cell", "Others"],
    "immune_extended": ["CD4 T cell", "CD8 T cell", "Dendritic cell", "B cell", "M1 macrophage cell", "M2 macrophage cell",
                        "Natural killer cell", "Others"],
    "immune_base": ["B cell", "CD4 T cell", "CD8 T cell", "Others", "Dendritic cell"],
    "struct": ["Stroma cell", "Smooth muscle", "Endothelial cell", "Epithelial cell", "Proliferating/tumor cell", "Others"],
    "nerve": ["Nerve cell", "Others"],
}
#: model name -> (parser panel name, checkpoint file under MODEL_DIR)
MODEL_PANEL = {"immune_base": "immune_base", "immune_extended": "immune_extended", "immune_full": "immune_full",
               "struct": "structure", "nerve": "nerve_cell"}
MODEL_DIR = "src/multiplexed_image_annotator/cell_type_annotation/models"   # CWD-relative, as in model.py:189-231
_GID = {name: i for i, name in enumerate(ops.GLOBAL_NAMES)}


class _LazyPredictions:
    """list of {cell type: probability} dicts (model.py:412-414) built on first access from the (n, K) table."""

    def __init__(self, model: str, probs: np.ndarray):
        self.model, self.probs, self._dicts = model, probs, None

    def _get(self):
        if self._dicts is None:
            names = CLASS_NAMES[self.model]
            self._dicts = [{names[i]: row[i] for i in range(len(row))} for row in self.probs]
        return self._dicts

    def __len__(self):
        return len(self.probs)

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())



def _load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """``{"model": state_dict}`` checkpoint as the reference stores it (model.py:189-231, markerImputer.py:260-271).

    The reference unpickles with ``weights_only=False``; a state dict is plain tensors, so the drop-in uses the loader that executes
    nothing from the file.  MAE-style training scripts (the lineage of the reference's checkpoints) save an ``args`` Namespace, an epoch
    number and optimizer / scaler state beside ``"model"``: ``argparse.Namespace`` is allow-listed for this one load -- rebuilding it
    runs no code from the file (its state is a dict the restricted unpickler has already vetted) -- so such a checkpoint loads as it does
    in the reference.  Anything else the restricted loader refuses is reported with the file name and the remedy; I/O errors (missing
    file, permissions, truncated archive) propagate as what they are.
    """
    import argparse
    import pickle
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except OSError:
        raise
    except (pickle.UnpicklingError, RuntimeError) as e:      # a global outside the allow-list: pickle.UnpicklingError (torch >= 2.4) / RuntimeError
        if isinstance(e, RuntimeError) and "weights_only" not in str(e).lower() and "unpickl" not in str(e).lower() and "global" not in str(e).lower():
            raise      # a truncated or corrupt archive, not a refused global: torch's own message says what is wrong
        raise RuntimeError(
            "{}: not a plain tensor checkpoint (torch.load(weights_only=True) refused it: {}). Re-save it as "
            "torch.save({{'model': model.state_dict()}}, path).".format(path, e)) from e
    if not isinstance(ckpt, dict) or "model" not in ckpt:
        raise RuntimeError("{}: expected a dict with a 'model' state dict".format(path))
    return ckpt["model"]


class Annotator(object):
    def __init__(self, marker_list_path, image_path, device, main_dir='./', batch_id='', strict=True, infer=True, min_cells=-1,
                 normalize=True, blur=False, amax=1, confidence=0.25, cell_size=30, cell_type_confidence=None, n_jobs=0):
        self.device = device
        self.cell_types = ["B cell", "CD4 T cell", "CD8 T cell", "Dendritic cell", "Regulatory T cell", "Granulocyte cell", "Mast cell",
                           "M1 macrophage cell", "M2 macrophage cell", "Natural killer cell", "Plasma cell", "Endothelial cell",
                           "Epithelial cell", "Stroma cell", "Smooth muscle", "Proliferating/tumor cell", "Nerve cell", "Others"]
        self.batch_id = batch_id
        self.rank, self.world_size = dist.world()
        log_dir = main_dir if self.rank == 0 else os.path.join(main_dir, f".rank{self.rank}")
        os.makedirs(log_dir, exist_ok=True)
        self.logger = Logger(log_dir)
        self.logger.log_all_hyperparameters({
            "Batch name": batch_id, "Strictly match panel(s)": strict, "Normalize image(s)": normalize,
            "Image blurring kernel size": blur, "Percentile of intensity to upper clip": amax, "Confidence threshold": confidence,
            "Estimated cell size (in pixels)": cell_size})
        self.logger.log("")
        self.logger.log("Start parsing the marker list.")
        self.channel_parser = MarkerParser(strict=strict, logger=self.logger)
        self.channel_parser.parse(marker_list_path)
        self.preprocessor = ImageProcessor(image_path, self.channel_parser, log_dir, device, batch_id, infer, normalize, blur, amax,
                                           cell_size, self.logger, n_jobs=n_jobs)
        # multi-rank runs: whole images per rank when the batch CSV has at least one per rank (reference main.py:39-52 batch_run; BASELINE
        # config 5: replicas only, nothing exchanged, every rank writes the CSVs of its own images under their batch-wide numbers), cells of
        # every image otherwise (contiguous shards, one all-gather per image, rank 0 writes)
        self.tile_mode = dist.tile_mode(self.preprocessor._n_images, self.world_size, os.environ.get("RIBCA_TILE_MODE"))
        self._loaded = False
        self.n_jobs = n_jobs
        self._n_images = 0
        self.min_cells = min_cells
        self.infer = infer
        self.annotations: List[List[str]] = []
        self.confidence: List[list] = []
        self.immune_annotations, self.struct_annotations, self.nerve_annotations = [], [], []
        self.immune_base_pred, self.immune_extended_pred, self.immune_full_pred = [], [], []
        self.struct_pred, self.nerve_pred = [], []
        self.confidence_thresh = confidence
        self.extra_cell_types = self.min_cells > 0
        if self.extra_cell_types:
            raise NotImplementedError("min_cells > 0 (UMAP/HDBSCAN re-clustering of 'Others', model.py:642-675) is post-analysis "
                                      "outside the accelerated hot path")
        self.n_regions = 0
        self.temp_dir = os.path.join(log_dir, "tmp")
        self.result_dir = os.path.join(main_dir, "results")
        os.makedirs(self.result_dir, exist_ok=True)
        if cell_type_confidence is None:
            self.cell_type_confidence = {name: -1 for name in self.cell_types}
        else:
            self.cell_type_confidence = cell_type_confidence
        self.models: Dict[str, ops.VitModel] = {}
        self.imputers: Dict[str, ops.MaeModel] = {}
        self._weights: Dict[str, Dict[str, torch.Tensor]] = {}
        self.probs: List[Dict[str, np.ndarray]] = []       # per image: model -> (n, K) fp32 host table
        self._conf_arrays: List[np.ndarray] = []           # per image: the float32 confidences behind self.confidence (-1.0 = thresholded)
        self.recheck_stats: List[Dict[str, int]] = []      # per image: cells, re-evaluated near a decision boundary, still within the noise floor
        self.label_ids: List[np.ndarray] = []
        self.chunk_cells = int(os.environ.get("RIBCA_CHUNK_CELLS", "1024"))
        self.streams = int(os.environ.get("RIBCA_STREAMS", "3"))      # cell segments in flight per classifier (ops.VitModel.predict_proba)

    # ---- weights ---------------------------------------------------------------------------------------------------
    def set_weights(self, weights: Dict[str, Dict[str, torch.Tensor]]) -> None:
        """Provide state dicts directly (timm key names) instead of the CWD-relative ``.pth`` files."""
        self._weights.update(weights)

    def load_models(self):
        """model.py:188-239: every checkpoint that exists is loaded; a missing one is reported and skipped."""
        dev = _lib.require_gpu()
        for name in ("immune_base", "immune_extended", "immune_full", "struct", "nerve"):
            sd = self._weights.get(name)
            path = os.path.join(MODEL_DIR, name + ".pth")
            if sd is None and os.path.exists(path):
                sd = _load_state_dict(path)
            if sd is None:
                msg = {"immune_base": "Immune base", "immune_extended": "Immune extended", "immune_full": "Immune full",
                       "struct": "Tissue structure", "nerve": "Nerve cell"}[name] + " model not found"
                print(msg)
                self.logger.log(msg)
                continue
            self.models[name] = ops.VitModel(sd, dev)
        self._loaded = True

    def _imputer(self, panel: str) -> "ops.MaeModel":
        """markerImputer.py:258-287: ``<panel>_impute.pth`` next to the classifier checkpoints, else ValueError("Panel not found")."""
        if panel not in self.imputers:
            sd = self._weights.get(panel + "_impute")
            path = os.path.join(MODEL_DIR, panel + "_impute.pth")
            if sd is None and panel in ("immune_full", "immune_extended", "immune_base") and os.path.exists(path):
                sd = _load_state_dict(path)
            if sd is None:
                raise ValueError("Panel not found")
            self.imputers[panel] = ops.MaeModel(sd, _lib.require_gpu())
            index = self.channel_parser.indices[panel]
            n_present = sum(1 for c in index if c != -1)
            # preprocess.py:272-279, including its loop bound (only the first len(present) entries are inspected)
            msg = "Imputer for {} is created. Marker(s) ".format(panel)
            for ii in range(n_present):
                if index[ii] == -1:
                    msg += "{} ".format(self.channel_parser.panels[panel][ii])
            msg += "are imputed."
            print("Imputer for {} is created".format(panel))
            self.logger.log(msg)
        return self.imputers[panel]

    def _panels_to_impute(self) -> List[str]:
        """preprocess.py:268: panels whose index list holds a -1, unless infer is off (never 'structure'; the reference's
        "nerve" test never matches the real panel name 'nerve_cell', whose panel tolerates no missing marker anyway)."""
        out = []
        for panel in self.channel_parser.panels:
            index = self.channel_parser.indices.get(panel)
            if index is not None and self.infer and -1 in index and panel not in ("structure", "nerve"):
                out.append(panel)
        return out

    # ---- pipeline --------------------------------------------------------------------------------------------------
    def preprocess(self):
        rank, ws = self.rank, self.world_size
        # The reference builds each MarkerImputer inside transform() (preprocess.py:272): a missing ``<panel>_impute.pth`` raises
        # ValueError("Panel not found") there, not in predict().  Here they are built up front, so that every rank fails before
        # any collective is entered (a rank raising later would leave the others waiting in the all-gather).
        for panel in self._panels_to_impute():
            self._imputer(panel)
        if self.tile_mode:
            self.preprocessor.transform(image_filter=lambda i: dist.owns_image(i, rank, ws))
            self._n_images = len(self.preprocessor.image_ids)      # every per-image list below is indexed by LOCAL position
            self.logger.log("rank {} of {}: tile-per-rank mode, images {} of {}".format(rank, ws, self.preprocessor.image_ids,
                                                                                        self.preprocessor._n_images))
            return
        if ws > 1 and os.environ.get("RIBCA_NORM_SHARD") == "1":
            self.preprocessor.norm_shard = (rank, ws)
        self.preprocessor.transform(shard_fn=(lambda n: dist.shard_bounds(n, rank, ws)) if ws > 1 else None,
                                    gather_fn=(lambda t, n: dist.all_gather_rows(t, n)) if ws > 1 else None)
        self._n_images = self.preprocessor._n_images

    def clear(self):
        self.immune_base_pred, self.immune_extended_pred, self.immune_full_pred = [], [], []
        self.struct_pred, self.nerve_pred = [], []
        self.annotations = []
        self._conf_arrays = []
        self.recheck_stats: List[Dict[str, int]] = []      # per image: cells, re-evaluated near a boundary, still within the noise floor

    def _active_models(self) -> Dict[str, Optional[str]]:
        """model.py:241-349: one immune model (full > extended > base) plus struct / nerve when their panels apply."""
        p = self.channel_parser
        immune = "immune_full" if p.immune_full else ("immune_extended" if p.immune_extended else ("immune_base" if p.immune_base else None))
        return {"immune": immune, "struct": "struct" if p.struct else None, "nerve": "nerve" if p.nerve else None}

    def _predict_cell_types(self, image_idx, model_name, batch_size=None, rows: Optional[torch.Tensor] = None, precise: bool = False) -> torch.Tensor:
        """softmax(model(x), dim=1) for THIS RANK's cells of one image: (n_local, K) device table (predict() gathers).  ``rows``: only
        these local cells (device index tensor); ``precise``: three fp16 passes per product whatever the width (the re-evaluation of
        cells near a decision boundary, see predict())."""
        pre = self.preprocessor
        model = self.models[model_name]
        index = self.channel_parser.indices[MODEL_PANEL[model_name]]
        c_img = pre.images_dev[image_idx].shape[0]
        src = ops.resolve_channels(index, c_img)
        patches = pre.panel_patches(image_idx)
        if rows is not None:
            patches = patches.index_select(0, rows)
        # preprocess.py:268-281: missing markers of an immune panel are imputed unless infer is off (never for structure / nerve)
        if self.infer and -1 in index and MODEL_PANEL[model_name] not in ("structure", "nerve"):
            imputer = self._imputer(MODEL_PANEL[model_name])
            sel = torch.tensor([max(c, 0) for c in src], dtype=torch.long, device=patches.device)
            panel = patches.index_select(1, sel).contiguous()           # channel gather (layout only); blanks are overwritten below
            present = [i for i, c in enumerate(index) if c != -1]
            imputer.impute(panel, present, chunk_cells=ops.MaeModel.CHUNK_FACTOR * self.chunk_cells)
            patches, src = panel, list(range(len(index)))
        if precise:
            return model._forward(patches, src, chunk_cells=self.chunk_cells, streams=1, precise=True)
        return model.predict_proba(patches, src, chunk_cells=self.chunk_cells, streams=self.streams)

    # cells whose vote lies this close to a decision boundary are re-evaluated with three fp16 passes per product (ops.VitModel.RECHECK_MARGIN);
    # those still within NOISE_FLOOR afterwards are counted and logged: two correct fp32 evaluations need not agree on their label
    NOISE_FLOOR = 2.0e-4

    def _recheck_near_boundaries(self, image_idx, tables, pair, tc) -> Dict[str, int]:
        """The MX arithmetic (csrc/gemm_mx.hip) moves softmax outputs by a few 1e-5: a cell whose vote sits within 1e-3 of a boundary
        (top-2 margin, "Others", confidence thresholds -- ops.decision_distance) is recomputed at the full 22-bit operand precision, for
        every model of the voting pair, and its table rows are replaced.  Returns the counts that predict() logs."""
        others = {k: (CLASS_NAMES[k].index("Others") if "Others" in CLASS_NAMES[k] else None) for k in pair if k}
        thresholds = [self.confidence_thresh] + [t for t in tc if t is not None and t >= 0]
        def distance():
            pb = tables[pair[1]] if pair[1] else None
            return ops.decision_distance(tables[pair[0]], others[pair[0]], pb, others.get(pair[1]) if pair[1] else None, thresholds)
        stats = {"cells": int(tables[pair[0]].shape[0]), "re_evaluated": 0, "within_noise_floor": 0}
        if stats["cells"] == 0:
            return stats
        d = distance()
        uses_mx = [k for k in pair if k and self.models[k].uses_mx]
        if uses_mx:
            margin = max(self.models[k].recheck_margin for k in uses_mx)
            stats["margin"] = margin
            rows = torch.nonzero(d < margin).flatten()
            if rows.numel():
                moved = 0.0
                for k in uses_mx:
                    again = self._predict_cell_types(image_idx, k, rows=rows, precise=True)
                    # how far the full-precision result lies from the fast one on these cells: the quantity RECHECK_MARGIN has to dominate
                    moved = max(moved, float((again - tables[k].index_select(0, rows)).abs().max().item()))
                    tables[k].index_copy_(0, rows, again)
                stats["re_evaluated"] = int(rows.numel())
                stats["max_fast_minus_full_precision"] = moved
                d = distance()
        stats["within_noise_floor"] = int((d < self.NOISE_FLOOR).sum().item())
        return stats

    def predict(self, batch_size=32):
        self.logger.log("\nStart predicting cell types and tissue structures.")
        if not self._loaded:
            self.load_models()
        active = self._active_models()
        for role, name in active.items():
            if name is not None and name not in self.models:
                raise AttributeError(f"'Annotator' object has no attribute '{name}_model'")   # what the reference ends up raising
        dev = _lib.require_gpu()
        tc = [self.cell_type_confidence[n] for n in ops.GLOBAL_NAMES]
        for image_idx in range(self._n_images):
            tables: Dict[str, torch.Tensor] = {}
            for role in ("immune", "struct", "nerve"):
                name = active[role]
                if name is None:
                    msg = {"immune": "No immune cell model to predict", "struct": "No structure model to predict",
                           "nerve": "No nerve cell model to predict"}[role]
                    print(msg)
                    self.logger.log(msg)
                    continue
                tables[name] = self._predict_cell_types(image_idx, name, batch_size)
            if not tables:
                raise ValueError("No predictions to merge")
            imm, st, nv = active["immune"], active["struct"], active["nerve"]
            if not (imm == "immune_full" and st and nv):      # (that combination raises below, as the reference does)
                pair0 = (imm, st) if imm and st else (st, nv) if st and nv else (imm, nv) if imm and nv else (imm or st or nv, None)
                rs = self._recheck_near_boundaries(image_idx, tables, pair0, tc)
                self.recheck_stats.append(rs)      # (this rank's shard: the counts are logged per rank, no collective for a log line)
                msg = ("{} of {} cells lay within {:g} of a decision boundary and were re-evaluated at full operand precision; {} remain within "
                       "{:g} (inside the arithmetic's noise floor: another correct fp32 evaluation may label them differently)."
                       ).format(rs["re_evaluated"], rs["cells"], rs.get("margin", ops.VitModel.RECHECK_MARGIN), rs["within_noise_floor"], self.NOISE_FLOOR)
                if "max_fast_minus_full_precision" in rs:
                    msg += " Largest move of a confidence under the re-evaluation: {:.1e} (margin {:g}).".format(rs["max_fast_minus_full_precision"],
                                                                                                                 rs.get("margin", ops.VitModel.RECHECK_MARGIN))
                self.logger.log(msg if self.world_size == 1 else "rank {}: {}".format(self.rank, msg))
            if self.world_size > 1 and not self.tile_mode:
                # ONE all-gather per image: the models' probability columns side by side (<= 33 floats per cell), SURVEY 8(e)
                names = list(tables)
                widths = [tables[k].shape[1] for k in names]
                full = dist.all_gather_rows(torch.cat([tables[k] for k in names], dim=1), len(self.preprocessor.cell_ids[image_idx]))
                tables = {k: t.contiguous() for k, t in zip(names, torch.split(full, widths, dim=1))}
            if imm == "immune_full" and st and nv:
                raise KeyError("Others")       # reference branch 1 (model.py:483-510) fails exactly like this
            if imm and st:
                pair = (imm, st)
            elif st and nv:
                pair = (st, nv)
            elif imm and nv:
                pair = (imm, nv)
            else:
                pair = (imm or st or nv, None)
            pa = tables[pair[0]]
            pb = tables[pair[1]] if pair[1] else None
            lab, conf = ops.vote(pa, [_GID[c] for c in CLASS_NAMES[pair[0]]], pb, [_GID[c] for c in CLASS_NAMES[pair[1]]] if pair[1] else None,
                                 tc, self.confidence_thresh)
            host = {k: v.cpu().numpy() for k, v in tables.items()}
            self.probs.append(host)
            for name, table in host.items():
                lazy = _LazyPredictions(name, table)
                if name.startswith("immune"):
                    getattr(self, name + "_pred").append(lazy)
                    self.immune_annotations.append(lazy)
                elif name == "struct":
                    self.struct_pred.append(lazy)
                    self.struct_annotations.append(lazy)
                else:
                    self.nerve_pred.append(lazy)
                    self.nerve_annotations.append(lazy)
            lab_h = lab.cpu().numpy().astype(np.int64)
            conf_h = conf.cpu().numpy()
            self.label_ids.append(lab_h)
            names = np.array(ops.GLOBAL_NAMES, dtype=object)
            self.annotations.append(names[lab_h].tolist())
            self.confidence.append([-1 if c == -1 else c for c in conf_h])    # int -1 marks a thresholded cell, as in model.py:507
            self._conf_arrays.append(conf_h)
        self.logger.log("Finished predicting cell types and tissue structures.")
        self.cell_types = self._get_unique_cell_types()
        self.cell_types = np.delete(self.cell_types, np.where(self.cell_types == "Others"))
        self.cell_types = np.append(self.cell_types, "Others")
        self.colors = colors.get_colors(len(self.cell_types))     # model.py:459 (the legend PNG of model.py:461-462 is not drawn)
        self._annotations_all = None

    def merge_by_voting(self):
        """model.py:481-640.  predict() has already voted (HIP vote kernel over the probability tables); calling this afterwards,
        as external code following the reference might, leaves the result as is.  Before predict() it fails as the reference does."""
        if len(self.annotations) == 0:
            raise ValueError("No predictions to merge")

    @property
    def annotations_all(self):
        """model.py:464-478, built on first access (it copies every cell's pixel lists)."""
        if getattr(self, "_annotations_all", None) is None:
            out = []
            for i in range(len(self.annotations)):
                pos = self.preprocessor.cell_pos_dict[i]
                rows = []
                for j, key in enumerate(pos.keys()):
                    cell_type_int = np.where(self.cell_types == self.annotations[i][j])[0][0]
                    r, c = pos[key]
                    rows.append({"Cell ID": key, "Cell type": cell_type_int, "Confidence": self.confidence[i][j], "Row": r, "Column": c})
                out.append(rows)
            self._annotations_all = out
        return self._annotations_all

    def _get_unique_cell_types(self):
        seen = set()
        for per_image in self.annotations:
            seen.update(per_image)
        if self.tile_mode:
            # the reference's list covers every image of the batch (model.py:678-686): the union over the ranks, as an 18-entry presence
            # vector (the only exchange of a tile-per-rank run: control plane, 144 bytes)
            present = torch.tensor([1 if name in seen else 0 for name in ops.GLOBAL_NAMES], dtype=torch.int64)
            present = dist.all_reduce_sum(present)
            seen = {name for name, p in zip(ops.GLOBAL_NAMES, present.tolist()) if p > 0}
        return np.sort(np.array(list(seen)))

    def get_cell_type_names(self):
        txt = ""
        for i in range(len(self.cell_types)):
            txt += f"{i+1}: {self.cell_types[i]}"
            txt += "\n" if i % 3 == 2 else "  "
        return txt

    def export_annotations(self):
        """model.py:768-795: same header, columns, rounding and number formatting."""
        if len(self.annotations) == 0:
            if self.tile_mode and self.preprocessor._n_images > 0:
                return       # a rank that owns no image of the batch has nothing to write
            raise ValueError("No annotations to export")
        if self.rank != 0 and not self.tile_mode:
            return
        for i in range(len(self.annotations)):
            path = os.path.join(self.result_dir, f"{self.batch_id}_annotation_{self._image_number(i)}.csv")
            ids = self.preprocessor.cell_ids[i]
            tab = self.preprocessor.cell_tables[i]
            conf = self.confidence[i]
            rows = np.round(tab[:, 4].astype(np.float64) / tab[:, 6].astype(np.float64), 2)
            cols = np.round(tab[:, 5].astype(np.float64) / tab[:, 6].astype(np.float64), 2)
            regions = getattr(self, "tissue_regions", None)
            # The reference prints ``round(np.float32, 3)`` through an f-string (the float32 widened to double, e.g.
            # 0.5360000133514404) and the int -1 of a thresholded cell as "-1".  Same text, rounded and widened in one numpy call
            # instead of 100 k Python round() calls when the confidences are still the float32 table predict() produced.
            arr = self._conf_arrays[i] if i < len(getattr(self, "_conf_arrays", [])) and len(self._conf_arrays[i]) == len(conf) else None
            # ``confidence`` is public state (the reference's own _find_extra_cell_types edits it after predict()): the cached table is
            # only used while it still says the same thing
            if arr is not None and not np.array_equal(arr, np.asarray(conf, dtype=np.float32)):
                arr = None
            if arr is not None:
                conf_txt = ["-1" if v == -1.0 else repr(v) for v in np.round(arr, 3).tolist()]
            else:
                conf_txt = [f"{round(c, 3)}" for c in conf]
            keys, labs, rl, cl = ids.tolist(), self.annotations[i], rows.tolist(), cols.tolist()
            with open(path, "w") as f:
                f.write("Cell Index,Cell Type,Confidence,Row,Column,Tissue Region\n")
                if regions is None:
                    f.write("".join([f"{k},{l},{c},{r},{cc},None\n" for k, l, c, r, cc in zip(keys, labs, conf_txt, rl, cl)]))
                else:
                    reg = regions[i]
                    f.write("".join([f"{k},{l},{c},{r},{cc},Region {reg[k]}\n" for k, l, c, r, cc in zip(keys, labs, conf_txt, rl, cl)]))
            self.logger.log(f"Exported annotations for image {i} to {path}")

    def min_cells_per_image(self) -> int:
        """fewest cells in any image of the batch (what the reference's k-NN calls need to exceed); in tile-per-rank mode the minimum over
        every rank's images, so that all ranks take the same branches of the pipeline (main._pipeline)"""
        n = min((len(ids) for ids in self.preprocessor.cell_ids), default=0)
        return dist.all_reduce_min_int(n) if self.tile_mode else n

    def _image_number(self, i: int) -> int:
        """row of the batch CSV that local position i holds: i itself except in tile-per-rank mode (file names carry the batch-wide number)"""
        ids = self.preprocessor.image_ids
        return ids[i] if i < len(ids) else i

    def _writes_files(self) -> bool:
        """rank 0 writes everything in cell-sharded runs (every rank holds the same tables after the all-gather); in tile-per-rank mode
        every rank writes the files of its own images"""
        return self.rank == 0 or self.tile_mode

    def clear_tmp(self):
        for f in os.listdir(self.temp_dir):
            os.remove(os.path.join(self.temp_dir, f))
        os.rmdir(self.temp_dir)
        self.logger.log("Temporary files cleared")

    # ---- label painting (model.py:806-858) -------------------------------------------------------------------------
    def paint(self, image_idx: int):
        """Device tensors (H, W, 3) uint8 cell-type colours, (H, W, 3) uint8 confidence colours (silver where thresholded),
        (H, W) uint8 cell-type index + 1 -- what ``colorize`` writes as PNGs.  One gather kernel per image instead of the
        reference's per-cell fancy indexing."""
        pre = self.preprocessor
        ids = pre.cell_ids[image_idx]
        types = {str(t): k for k, t in enumerate(self.cell_types)}
        gid_to_type = np.array([types.get(name, 0) for name in ops.GLOBAL_NAMES], dtype=np.int64)
        tidx = gid_to_type[self.label_ids[image_idx]]
        palette = np.array(self.colors, dtype=np.uint8)
        conf = np.array([float(c) for c in self.confidence[image_idx]], dtype=np.float32)
        return ops.colorize(pre.masks_dev[image_idx], ids, palette[tidx], colors.confidence_colors(conf), (tidx + 1).astype(np.uint8))

    def colorize(self, from_script=False):
        if len(self.preprocessor.masks) == 0:
            raise ValueError("No masks to colorize")
        if len(self.annotations) == 0:
            raise ValueError("No annotations to colorize")
        from PIL import Image
        for i in range(len(self.preprocessor.masks)):
            type_rgb, conf_rgb, type_idx = (t.cpu().numpy() for t in self.paint(i))
            if not self._writes_files():
                continue
            num = self._image_number(i)
            Image.fromarray(type_rgb).save(os.path.join(self.result_dir, f"{self.batch_id}_colorized_annotation_{num}.png"))
            if not from_script:            # napari working file of the reference GUI (model.py:845-847), only inside its source tree
                gui_dir = "./src/multiplexed_image_annotator/cell_type_annotation/_working_dir_temp"
                if os.path.isdir(gui_dir):
                    Image.fromarray(type_idx).save(os.path.join(gui_dir, "output_img.png"))
            Image.fromarray(conf_rgb).save(os.path.join(self.result_dir, f"{self.batch_id}_confidence_{num}.png"))
            if self.n_regions > 0:             # model.py:823-855: region colours from the same palette, silver last
                pre = self.preprocessor
                ids = pre.cell_ids[i]
                region = np.array([self.tissue_regions[i][int(k)] for k in ids.tolist()], dtype=np.int64)
                palette = np.array(colors.get_colors(self.n_regions + 1), dtype=np.uint8)
                t_rgb, _, t_idx = ops.colorize(pre.masks_dev[i], ids, palette[region], palette[region], (region + 1).astype(np.uint8))
                Image.fromarray(t_rgb.cpu().numpy()).save(os.path.join(self.result_dir, f"{self.batch_id}_tissue_region_{num}.png"))
                if not from_script and os.path.isdir("./src/multiplexed_image_annotator/cell_type_annotation/_working_dir_temp"):
                    Image.fromarray(t_idx.cpu().numpy()).save("./src/multiplexed_image_annotator/cell_type_annotation/_working_dir_temp/output_img_2.png")

    # ---- neighbourhood analysis (model.py:798-800 -> spatial_methods.py:13-130) -----------------------------------------
    def _cell_type_ints(self, image_idx: int) -> np.ndarray:
        types = {str(t): k for k, t in enumerate(self.cell_types)}
        gid_to_type = np.array([types.get(name, 0) for name in ops.GLOBAL_NAMES], dtype=np.int64)
        return gid_to_type[self.label_ids[image_idx]]

    def neighborhood_matrix(self, image_indices, n_neighbors=25) -> np.ndarray:
        """Counts of (cell type, neighbour cell type) over every cell's n_neighbors - 1 nearest other cells, summed over the images."""
        t = len(self.cell_types)
        acc = None
        for i in image_indices:
            tab = self.preprocessor.cell_tables[i]
            x = tab[:, 5].astype(np.float64) / tab[:, 6].astype(np.float64)      # np.mean(Column), np.mean(Row) of the reference
            y = tab[:, 4].astype(np.float64) / tab[:, 6].astype(np.float64)
            acc = ops.knn_cooccurrence(x, y, self._cell_type_ints(i), t, n_neighbors, out=acc)
        if acc is None:      # a rank that owns no image of the batch (tile-per-rank mode forced on a batch smaller than the world)
            return np.zeros((t, t), dtype=np.float64)
        return acc.cpu().numpy().astype(np.float64)

    def neighborhood_analysis(self, n_neighbors=25, integrate=True, normalize=True):
        """Writes the same CSVs as the reference (``{batch_id}_integrated_neighborhood.csv`` or one ``{batch_id}_neighborhood_{i}.csv``
        per image); the seaborn heat-map PNGs are not drawn."""
        if len(self.annotations) == 0:
            raise ValueError("No annotations")
        groups = [list(range(self._n_images))] if integrate else [[i] for i in range(self._n_images)]
        for g, idx in enumerate(groups):
            m = self.neighborhood_matrix(idx, n_neighbors)
            if integrate and self.tile_mode:
                # the integrated matrix sums over ALL images of the batch: T x T counts from every rank (cell_types is the batch-wide union on
                # every rank, _get_unique_cell_types, so the type axes agree)
                m = dist.all_reduce_sum(torch.from_numpy(m)).numpy()
            if normalize:
                sums = m.sum(axis=1, keepdims=True)
                m = np.divide(m, sums, out=m.copy(), where=sums > 0)
            name = f"{self.batch_id}_integrated_neighborhood.csv" if integrate else f"{self.batch_id}_neighborhood_{self._image_number(g)}.csv"
            if self.rank == 0 or (self.tile_mode and not integrate):
                with open(os.path.join(self.result_dir, name), "w") as f:
                    f.write("cell_type," + "".join(f"{c}," for c in self.cell_types) + "\n")
                    for r, c in enumerate(self.cell_types):
                        f.write(f"{c}," + "".join(f"{m[r][j]:.3f}," for j in range(len(self.cell_types))) + "\n")

    # ---- tissue regions (model.py:802-804 -> spatial_methods.py:133-198) -------------------------------------------------
    def tissue_region_analysis(self, n, method="kmeans"):
        """Per-cell region labels from the cell-type make-up of each cell's 10 ... 200 nearest neighbours.  The 201-NN search and
        the counting run on the GPU (ops.knn_compositions); PCA(0.99) and the clustering are the same scikit-learn calls as in
        the reference (their random start is not seeded there either, so labels are reproducible only up to that)."""
        from sklearn.cluster import HDBSCAN, KMeans, SpectralClustering
        from sklearn.decomposition import PCA
        self.n_regions = n
        self.tissue_regions = []
        for i in range(self._n_images):
            tab = self.preprocessor.cell_tables[i]
            x = tab[:, 5].astype(np.float64) / tab[:, 6].astype(np.float64)
            y = tab[:, 4].astype(np.float64) / tab[:, 6].astype(np.float64)
            types = self._cell_type_ints(i)
            comp = ops.knn_compositions(x, y, types, int(types.max()) + 1)
            comp = PCA(n_components=0.99).fit_transform(comp)
            if method == "kmeans":
                clusterer = KMeans(n_clusters=n)
            elif method == "hdbscan":
                clusterer = HDBSCAN(n_clusters=n)        # as written in the reference (raises there too: HDBSCAN has no n_clusters)
            elif method == "spectral":
                clusterer = SpectralClustering(n_clusters=n, n_jobs=self.n_jobs if self.n_jobs and self.n_jobs > 0 else None)
            else:
                raise UnboundLocalError("local variable 'clusterer' referenced before assignment")
            labels = clusterer.fit_predict(comp)
            self.tissue_regions.append({int(k): labels[j] for j, k in enumerate(self.preprocessor.cell_ids[i].tolist())})

    # ---- outside the accelerated path ------------------------------------------------------------------------------
    def _skip(self, what: str):
        msg = f"{what}: skipped (plotting downstream of the CSV, outside the accelerated hot path)"
        self.logger.log(msg)
        return None

    def generate_heatmap(self, integrate=False):
        """model.py:697-766 (seaborn heat-maps of the intensity table): not drawn; ``preprocessor.intensity_full`` holds the data."""
        return self._skip("generate_heatmap")

    def cell_type_composition(self, reduction=True):
        """model.py:860-913 (pie charts): not drawn; the CSVs hold the labels."""
        return self._skip("cell_type_composition")

    def umap_visualization(self, *_a, **_k):
        return self._skip("umap_visualization")
