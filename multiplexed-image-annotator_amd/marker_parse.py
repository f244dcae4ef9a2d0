"""Marker-name -> panel channel indices: host-side boundary logic of the hot path, same public surface as the
reference's ``MarkerParser`` (cell_type_annotation/markerParse.py:4-117): ``panels``, ``indices``, ``markers``,
``n_markers``, the five applicability flags and ``parse(marker_file)``.  Table-driven; the matching rules are

* exact string match after one alias pass (an alias is applied only if its target is not already in the list);
* non-strict mode tolerates up to ``MISSING_ALLOWED[panel]`` absent markers (index -1) in panels longer than 3;
* the marker file is read with ``np.loadtxt(dtype=str)`` like the reference, so alias replacement inherits numpy's
  fixed-width string truncation (a replacement longer than the widest name in the file is cut: 'CK' -> 'PanC').
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

PANELS: Dict[str, List[str]] = {
    "immune_base": ['CD45', 'CD20', 'CD4', 'CD8', 'DAPI', 'CD11c', 'CD3'],
    "immune_extended": ['DAPI', 'CD3', 'CD4', 'CD8', 'CD11c', 'CD20', 'CD45', 'CD68', 'CD163', 'CD56'],
    "immune_full": ['DAPI', 'CD3', 'CD4', 'CD8', 'CD11c', 'CD15', 'CD20', 'CD45', 'CD56', 'CD68', 'CD138', 'CD163', 'FoxP3',
                    'Granzyme B', 'Trypase'],   # sic: the reference spells it 'Trypase' (markerParse.py:13)
    "structure": ['DAPI', 'aSMA', 'CD31', 'PanCK', 'Vimentin', 'Ki67', 'CD45'],
    "nerve_cell": ['DAPI', 'CD45', 'GFAP'],
}
MISSING_ALLOWED = {"immune_base": 1, "immune_extended": 2, "immune_full": 3, "structure": 1, "nerve_cell": 0}
ALIASES = {'DNA': 'DAPI', 'DPAI-02': 'DAPI', 'CD16': 'CD15', 'CD38': 'CD138', 'CD79': 'CD20', 'CHGA': 'GFAP', 'SMActin': 'aSMA',
           'CD3e': 'CD3', 'CK': 'PanCK', 'CytoKeratin': 'PanCK', 'Cytokeratin': 'PanCK', 'Cytokeratin-19': 'PanCK', 'panCK': 'PanCK'}
ALTERNATIVES = {"CD20": "CD20 or CD79a", "GFAP": "GFAP or Chromogranin A", "CD138": "CD138 or CD38"}


class MarkerParser:
    def __init__(self, strict=True, logger=None):
        self.panels = {name: list(markers) for name, markers in PANELS.items()}
        self.indices: Dict[str, Optional[List[int]]] = {}
        self.immune_base = False
        self.immune_extended = False
        self.immune_full = False
        self.struct = False
        self.nerve = False
        self.strict = strict
        self.markers: List[str] = []
        self.logger = logger

    def _say(self, text: str) -> None:
        if self.logger:
            self.logger.log(text)

    def _matching(self, marker_list, panel, panel_name):
        found: List[int] = []
        absent: List[str] = []
        tolerant = (not self.strict) and len(panel) > 3
        for marker in panel:
            if marker in marker_list:
                found.append(marker_list.index(marker))
                continue
            shown = ALTERNATIVES.get(marker, marker)
            if not tolerant:
                print(f"Marker {shown} is not found in the list, ", end="")
                self._say(f"Marker {shown} is not found in the list.")
                return None
            absent.append(shown)
            found.append(-1)
            if len(absent) > MISSING_ALLOWED[panel_name]:
                joined = ', '.join(absent)
                print(f"Markers {joined} are not found in the list, ", end="")
                self._say(f"Markers {joined} are not found in the list.")
                return None
        return found

    def parse(self, marker_file):
        names = np.loadtxt(marker_file, delimiter=',', dtype=str)
        self.markers.extend(names)
        self._say("The panel contains the following markers: " + ", ".join(str(m) for m in names) + ".")
        for i in range(len(names)):
            target = ALIASES.get(str(names[i]))
            if target is not None and target not in names:
                before = names[i]
                names[i] = target          # fixed-width unicode array: may truncate, as in the reference
                self._say(f"Replaced the marker name {before} with {names[i]} to match our panel.")
        self._say("")
        marker_list = list(names)
        self.n_markers = len(marker_list)
        for panel_name, panel in self.panels.items():
            matched = self._matching(marker_list, panel, panel_name)
            applied = bool(matched)
            self.indices[panel_name] = matched if applied else None
            verdict = f"{panel_name} panel is applied." if applied else f"{panel_name} panel is not applied."
            print(verdict)
            self._say(verdict)
            self._say("\n")
        self.immune_base = bool(self.indices['immune_base'])
        self.immune_extended = bool(self.indices['immune_extended'])
        self.immune_full = bool(self.indices['immune_full'])
        self.struct = bool(self.indices['structure'])
        self.nerve = bool(self.indices['nerve_cell'])
