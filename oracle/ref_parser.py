"""ORACLE (test infrastructure).  Restatement of the reference's marker-panel matching
(``cell_type_annotation/markerParse.py:4-117``), pinned by tests/golden/parser_cases.json which was
produced by the reference's own ``MarkerParser`` (pure Python + numpy, runs unmodified here).

Returned value mirrors what the hot path reads from the reference object: ``indices[panel]`` (list with
-1 for tolerated missing markers, or None), the five applicability flags and the marker list as read.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

#: markerParse.py:8-17 (note the literal 'Trypase'); dict order = panel order
PANEL_MARKERS: Dict[str, List[str]] = {
    "immune_base": ['CD45', 'CD20', 'CD4', 'CD8', 'DAPI', 'CD11c', 'CD3'],
    "immune_extended": ['DAPI', 'CD3', 'CD4', 'CD8', 'CD11c', 'CD20', 'CD45', 'CD68', 'CD163', 'CD56'],
    "immune_full": ['DAPI', 'CD3', 'CD4', 'CD8', 'CD11c', 'CD15', 'CD20', 'CD45', 'CD56', 'CD68', 'CD138', 'CD163',
                    'FoxP3', 'Granzyme B', 'Trypase'],
    "structure": ['DAPI', 'aSMA', 'CD31', 'PanCK', 'Vimentin', 'Ki67', 'CD45'],
    "nerve_cell": ['DAPI', 'CD45', 'GFAP'],
}
#: markerParse.py:33
MISSING_ALLOWED = {"immune_base": 1, "immune_extended": 2, "immune_full": 3, "structure": 1, "nerve_cell": 0}
#: markerParse.py:76-77
ALIASES = {'DNA': 'DAPI', 'DPAI-02': 'DAPI', 'CD16': 'CD15', 'CD38': 'CD138', 'CD79': 'CD20', 'CHGA': 'GFAP',
           'SMActin': 'aSMA', 'CD3e': 'CD3', 'CK': 'PanCK', 'CytoKeratin': 'PanCK', 'Cytokeratin': 'PanCK',
           'Cytokeratin-19': 'PanCK', 'panCK': 'PanCK'}


def match_panel(names: List[str], wanted: List[str], panel: str, strict: bool) -> Optional[List[int]]:
    """markerParse.py:30-60.  Exact-string lookup; a miss is tolerated (index -1) only in non-strict mode
    for panels longer than 3 markers and while the number of misses does not exceed the panel's allowance."""
    hits: List[int] = []
    misses = 0
    for m in wanted:
        if m in names:
            hits.append(names.index(m))
            continue
        if strict or len(wanted) <= 3:
            return None
        misses += 1
        hits.append(-1)
        if misses > MISSING_ALLOWED[panel]:
            return None
    return hits


def parse_marker_file(path: str, strict: bool = True) -> dict:
    """markerParse.py:62-117.  ``np.loadtxt(dtype=str)`` yields a fixed-width unicode array, so an alias
    replacement longer than the widest name in the file is silently truncated (e.g. 'CK' -> 'PanC' in a file
    whose longest name has 4 characters) -- reproduced by doing the replacement on the same array type."""
    arr = np.loadtxt(path, delimiter=',', dtype=str)
    markers = [str(m) for m in arr]
    for i in range(len(arr)):
        if arr[i] in ALIASES and ALIASES[arr[i]] not in arr:
            arr[i] = ALIASES[arr[i]]
    names = list(arr)
    indices: Dict[str, Optional[List[int]]] = {}
    for panel, wanted in PANEL_MARKERS.items():
        got = match_panel(names, wanted, panel, strict)
        indices[panel] = got if got else None
    return {
        "markers": markers,
        "n_markers": len(names),
        "indices": indices,
        "immune_base": bool(indices["immune_base"]),
        "immune_extended": bool(indices["immune_extended"]),
        "immune_full": bool(indices["immune_full"]),
        "struct": bool(indices["structure"]),
        "nerve": bool(indices["nerve_cell"]),
    }
