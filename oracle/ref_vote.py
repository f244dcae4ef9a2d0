"""ORACLE (test infrastructure).  Restatement of the reference's per-cell vote, label table and CSV writer:

* ``CLASS_NAMES``           <- per-model index->name maps, ``model.py:247-252, 266-270, 284-287, 309-312, 334``
* ``VOTE_ORDER``            <- ``utils.py:143-146`` (get_void_vote key order = tie-break order of ``max``)
* ``merge_by_voting``       <- ``model.py:481-636``
* ``unique_cell_types``     <- ``model.py:455-458, 678-686``
* ``annotation_csv``        <- ``model.py:768-795``

Pinned by tests/golden/vote_cases.npz, produced by the reference's own ``Annotator.merge_by_voting`` /
``predict`` tail / ``export_annotations`` run on seeded probability tables in the build container.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

CLASS_NAMES: Dict[str, List[str]] = {
    "immune_full": ["CD4 T cell", "CD8 T cell", "Dendritic cell", "B cell", "M1 macrophage cell", "M2 macrophage cell",
                    "Regulatory T cell", "Granulocyte cell", "Plasma cell", "Natural killer cell", "Mast cell", "Others"],
    "immune_extended": ["CD4 T cell", "CD8 T cell", "Dendritic cell", "B cell", "M1 macrophage cell", "M2 macrophage cell",
                        "Natural killer cell", "Others"],
    "immune_base": ["B cell", "CD4 T cell", "CD8 T cell", "Others", "Dendritic cell"],
    "struct": ["Stroma cell", "Smooth muscle", "Endothelial cell", "Epithelial cell", "Proliferating/tumor cell", "Others"],
    "nerve": ["Nerve cell", "Others"],
}

VOTE_ORDER: List[str] = ["CD4 T cell", "CD8 T cell", "Dendritic cell", "B cell", "M1 macrophage cell", "M2 macrophage cell",
                         "Regulatory T cell", "Granulocyte cell", "Plasma cell", "Natural killer cell", "Mast cell",
                         "Stroma cell", "Smooth muscle", "Endothelial cell", "Epithelial cell", "Proliferating/tumor cell",
                         "Nerve cell"]

DEFAULT_TYPE_CONF = {name: -1 for name in VOTE_ORDER + ["Others"]}


def probs_to_dicts(model: str, probs: np.ndarray) -> List[Dict[str, np.float32]]:
    """model.py:412-414: one {cell type: probability} dict per cell, in class-index order."""
    names = CLASS_NAMES[model]
    return [{names[i]: row[i] for i in range(len(row))} for row in probs]


def _two_way(a: List[dict], b: List[dict], conf_thresh, type_conf) -> Tuple[list, list]:
    """model.py:512-591 (branches 2-4 share this body)."""
    labels, confs = [], []
    for pa, pb in zip(a, b):
        vote = {k: 0 for k in VOTE_ORDER}
        for pred in (pa, pb):
            for k in pred:
                if k != "Others":
                    vote[k] += pred[k]
        o1, o2 = pa["Others"], pb["Others"]
        best = max(vote, key=vote.get)
        thresh = min(o1, o2, conf_thresh) if type_conf[best] < 0 else type_conf[best]
        if vote[best] < thresh:
            labels.append("Others")
            confs.append(-1)
        else:
            labels.append(best)
            confs.append(vote[best])
    return labels, confs


def _one_way(a: List[dict], conf_thresh, type_conf) -> Tuple[list, list]:
    """model.py:593-633 (branches 5-7)."""
    labels, confs = [], []
    for pred in a:
        best = max(pred, key=pred.get)
        thresh = type_conf[best] if type_conf[best] > 0 else conf_thresh
        if best != "Others" and pred[best] < thresh:
            labels.append("Others")
            confs.append(-1)
        else:
            labels.append(best)
            confs.append(pred[best])
    return labels, confs


def merge_by_voting(immune: Optional[List[dict]], immune_kind: Optional[str], struct: Optional[List[dict]],
                    nerve: Optional[List[dict]], conf_thresh=0.25, type_conf: Optional[dict] = None) -> Tuple[list, list]:
    """model.py:481-636 for one image.  ``immune_kind`` in {immune_full, immune_extended, immune_base}.
    Branch 1 (immune_full + struct + nerve) adds the "Others" probability into a vote dict that has no such
    key and therefore raises KeyError('Others') in the reference; reproduced."""
    type_conf = DEFAULT_TYPE_CONF if type_conf is None else type_conf
    if immune is not None and immune_kind == "immune_full" and struct is not None and nerve is not None:
        raise KeyError("Others")
    if immune is not None and struct is not None:
        return _two_way(immune, struct, conf_thresh, type_conf)
    if struct is not None and nerve is not None:
        return _two_way(struct, nerve, conf_thresh, type_conf)
    if immune is not None and nerve is not None:
        return _two_way(immune, nerve, conf_thresh, type_conf)
    if immune is not None:
        return _one_way(immune, conf_thresh, type_conf)
    if struct is not None:
        return _one_way(struct, conf_thresh, type_conf)
    if nerve is not None:
        return _one_way(nerve, conf_thresh, type_conf)
    raise ValueError("No predictions to merge")


def unique_cell_types(annotations: Sequence[Sequence[str]]) -> np.ndarray:
    """model.py:678-686 then 455-458: sorted unique labels over all images, "Others" moved to the end
    (appended even when no cell is labelled Others)."""
    seen = set()
    for per_image in annotations:
        seen.update(per_image)
    types = np.sort(np.array(list(seen)))
    types = np.delete(types, np.where(types == "Others"))
    return np.append(types, "Others")


def annotation_csv(cell_ids: Sequence[int], labels: Sequence[str], confs: Sequence, sum_r: Sequence[int], sum_c: Sequence[int],
                   count: Sequence[int], regions: Optional[dict] = None) -> str:
    """model.py:768-795.  ``np.mean`` of an integer pixel list equals sum/count exactly in fp64 (integer sums
    below 2^53), so the per-cell sums stand in for the pixel lists."""
    out = ["Cell Index,Cell Type,Confidence,Row,Column,Tissue Region\n"]
    for j, key in enumerate(cell_ids):
        conf = round(confs[j], 3)
        row = round(np.float64(sum_r[j]) / np.float64(count[j]), 2)
        col = round(np.float64(sum_c[j]) / np.float64(count[j]), 2)
        region = "Region " + str(regions[key]) if regions is not None else None
        out.append(f"{key},{labels[j]},{conf},{row},{col},{region}\n")
    return "".join(out)
