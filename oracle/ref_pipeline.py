"""ORACLE (test infrastructure).  End-to-end CPU restatement of the reference hot path for one image:
``Annotator.preprocess() -> predict() -> export_annotations()`` (``cell_type_annotation/model.py:169-171,
431-478, 768-795`` and ``preprocess.py:241-290``), assembled from the per-stage restatements.  Marker
imputation (infer=True with missing markers) is handled by ``ref_mae`` when weights are supplied.

Pinned by tests/golden/e2e.{json,npz}, produced by the reference's own ``Annotator`` in the build container.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import ref_mae, ref_parser, ref_preprocess as rp, ref_vit, ref_vote

#: panel name (parser) -> model name, in the reference's panel iteration order (markerParse.py:8-17)
PANEL_MODEL = {"immune_base": "immune_base", "immune_extended": "immune_extended", "immune_full": "immune_full",
               "structure": "struct", "nerve_cell": "nerve"}


def choose_models(parsed: dict) -> Dict[str, Optional[str]]:
    """model.py:241-349: one immune model (full > extended > base), plus struct and nerve when applicable."""
    immune = None
    if parsed["immune_full"]:
        immune = "immune_full"
    elif parsed["immune_extended"]:
        immune = "immune_extended"
    elif parsed["immune_base"]:
        immune = "immune_base"
    return {"immune": immune, "struct": "struct" if parsed["struct"] else None, "nerve": "nerve" if parsed["nerve"] else None}


def run_image(raw: np.ndarray, mask: np.ndarray, marker_file: str, weights: Dict[str, Dict[str, torch.Tensor]], strict=True,
              normalize=True, blur=0, amax=100, confidence=0.25, type_conf=None, batch_size=32, cell_size=30, infer=False,
              imputers: Optional[Dict[str, Dict[str, torch.Tensor]]] = None) -> dict:
    parsed = ref_parser.parse_marker_file(marker_file, strict=strict)
    mask = np.asarray(mask)
    if mask.ndim == 3:
        mask = mask[:, :, 0]
    mask = mask.astype(np.int32)
    image = rp.normalize_image(raw, blur=blur, amax=amax) if normalize else raw
    ids, table = rp.cell_table(mask)
    which = choose_models(parsed)
    inv_panel = {v: k for k, v in PANEL_MODEL.items()}
    probs: Dict[str, np.ndarray] = {}
    intensity = None
    first = True
    patches_by_model = {}
    for panel in ref_parser.PANEL_MARKERS:            # preprocess.py:260-289: every applicable panel is cropped
        index = parsed["indices"][panel]
        if index is None:
            continue
        patches, inten = rp.patches_for_panel(image, mask, index, ids, table, scale=cell_size / 30.0, want_intensity=first)
        if first:
            intensity = inten
            first = False
        if infer and -1 in index and panel not in ("structure", "nerve"):          # preprocess.py:268-281
            if imputers is None or panel not in imputers:
                raise ValueError("Panel not found")                                 # markerImputer.py:276
            present = [i for i, c in enumerate(index) if c != -1]
            patches = ref_mae.impute(imputers[panel], torch.from_numpy(patches), present).numpy()
        patches_by_model[PANEL_MODEL[panel]] = patches
    for role, model in which.items():
        if model is None:
            continue
        probs[model] = ref_vit.predict_proba(weights[model], torch.from_numpy(patches_by_model[model]), batch_size).numpy()
    dicts = {m: ref_vote.probs_to_dicts(m, p) for m, p in probs.items()}
    labels, confs = ref_vote.merge_by_voting(dicts.get(which["immune"]) if which["immune"] else None, which["immune"],
                                             dicts.get("struct") if which["struct"] else None,
                                             dicts.get("nerve") if which["nerve"] else None, confidence, type_conf)
    cell_types = ref_vote.unique_cell_types([labels])
    csv = ref_vote.annotation_csv(ids.tolist(), labels, confs, table[:, 4], table[:, 5], table[:, 6])
    return {"parsed": parsed, "ids": ids, "table": table, "image": image, "patches": patches_by_model, "probs": probs,
            "labels": labels, "conf": confs, "cell_types": cell_types, "csv": csv, "intensity": intensity,
            "type_ints": [int(np.where(cell_types == l)[0][0]) for l in labels]}
