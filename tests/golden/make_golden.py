#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python functions.

Run in the build container only (``python tests/golden/make_golden.py``); /root/reference does not exist on
the GPU box and nothing in tests/, bench.py or __graft_entry__.py reads it.  Only data (inputs, expected
outputs) is written -- never reference source text.

How the reference is imported: ``cell_type_annotation/__init__.py`` is empty, so its modules are loaded as
the synthetic package ``refcta`` whose ``__path__`` is that directory (this bypasses the napari-widget import
of the parent package, which fails with an ordinary ModuleNotFoundError: magicgui).  Third-party modules
that are not installed get thin stand-ins registered in ``sys.modules`` first:

* ``skimage.{io,morphology,filters,transform}`` -> oracle/skimage_like.py (scipy.ndimage-backed restatement)
* ``timm.models.vision_transformer``            -> tests/golden/timm_like.py (plain-torch restatement)
* ``tifffile``, ``seaborn``, ``umap``           -> empty modules (plot/IO helpers, never called here)

Consequently: fixtures whose arithmetic is pure reference + numpy/scipy/torch (normalize, cell positions,
crop window/padding/channel-select control flow, parser, vote, CSV) pin the oracle to the reference; the
soft-mask filters and the ViT forward additionally pass through the restated third-party semantics.
"""
from __future__ import annotations

import hashlib
import importlib
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

REF = "/root/reference"
CTA = os.path.join(REF, "src/multiplexed_image_annotator/cell_type_annotation")

from multiplexed_image_annotator_amd import synth  # noqa: E402
from oracle import skimage_like  # noqa: E402
import timm_like  # noqa: E402


def install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    def imread(path):
        if str(path).endswith(".npy"):
            return np.load(path)
        from PIL import Image
        return np.array(Image.open(path))

    sk = mod("skimage")
    sk.io = mod("skimage.io", imread=imread)
    sk.morphology = mod("skimage.morphology", dilation=skimage_like.dilation, disk=skimage_like.disk)
    sk.filters = mod("skimage.filters", gaussian=skimage_like.gaussian)
    sk.transform = mod("skimage.transform", resize=skimage_like.resize)
    mod("tifffile", imwrite=lambda *a, **k: None)
    mod("seaborn", heatmap=lambda *a, **k: None)
    mod("umap")
    tm = mod("timm")
    tm.models = mod("timm.models")
    tm.models.vision_transformer = mod("timm.models.vision_transformer", VisionTransformer=timm_like.VisionTransformer,
                                       PatchEmbed=timm_like.PatchEmbed, Block=timm_like.Block)
    import matplotlib
    matplotlib.use("Agg")
    pkg = types.ModuleType("refcta")
    pkg.__path__ = [CTA]
    sys.modules["refcta"] = pkg


def ref(name):
    return importlib.import_module("refcta." + name)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class _Log:
    def log(self, *_a, **_k):
        pass

    def log_all_hyperparameters(self, *_a, **_k):
        pass


# ---------------------------------------------------------------------------------------------- G1
def small_raw_tile(c, h, w, seed):
    """uint16 test tile: synthetic cells plus deliberately degenerate channels."""
    _, img = synth.make_mask_and_image(h, w, max(4, (h * w) // 500), c, seed)
    return img.numpy().astype(np.uint16)


def golden_normalize():
    pre = ref("preprocess")
    dummy = object.__new__(pre.ImageProcessor)
    out = {}
    tile = small_raw_tile(4, 96, 112, synth.SEED_BASE + 101)
    tile[2] = 0                       # no positive pixel after background subtraction -> all -1
    tile[3] = (tile[3] % 23)          # dim channel: percentile <= 20 and max < 25 branches
    out["a_in"] = tile
    for blur in (0, 0.3, 0.5, 1):
        out[f"a_out_blur{blur}"] = pre.ImageProcessor._normalize(dummy, tile.copy(), blur=blur, amax=99.8)
    out["a_out_amax100"] = pre.ImageProcessor._normalize(dummy, tile.copy(), blur=0, amax=100)
    tile2 = small_raw_tile(3, 200, 168, synth.SEED_BASE + 102)   # larger than the sigma=20 radius (80) in both axes
    out["b_in"] = tile2
    out["b_out_blur0.3"] = pre.ImageProcessor._normalize(dummy, tile2.copy(), blur=0.3, amax=99.8)
    tile3 = small_raw_tile(2, 40, 56, synth.SEED_BASE + 103)     # smaller than the filter radius: multiple reflections
    out["c_in"] = tile3
    out["c_out_blur0"] = pre.ImageProcessor._normalize(dummy, tile3.copy(), blur=0, amax=99.8)
    np.savez_compressed(os.path.join(HERE, "normalize.npz"), **out)
    print("normalize.npz", {k: v.shape for k, v in out.items()})


# ---------------------------------------------------------------------------------------------- G2
def odd_mask():
    """Hand-built mask: border-touching cells, a >40 px cell, 1-px cells, non-contiguous ids, a cell in two pieces."""
    m = np.zeros((90, 120), np.int32)
    m[0:5, 0:7] = 3            # top-left corner
    m[85:90, 110:120] = 7      # bottom-right corner
    m[0:3, 50:60] = 12         # top edge
    m[40:50, 0:4] = 500        # left edge, id gap
    m[20:75, 30:95] = 65000    # 55 x 65 px: larger than the 40 px window
    m[30:40, 40:50] = 41       # cell nested inside the big one
    m[10, 100] = 9             # single pixel
    m[89, 0] = 10              # single pixel in the bottom-left corner
    m[60:63, 100:103] = 77     # two disjoint pieces of one id
    m[70:72, 112:118] = 77
    m[5:12, 110:120] = 100000  # right edge, id above uint16
    return m


def table_from_dict(d):
    ids = np.array(list(d.keys()), np.int64)
    tab = np.array([[min(r), max(r), min(c), max(c), sum(r), sum(c), len(r)] for r, c in d.values()], np.int64)
    return ids, tab


def golden_cellpos():
    pre = ref("preprocess")
    dummy = object.__new__(pre.ImageProcessor)
    out = {}
    m = odd_mask()
    d = pre.ImageProcessor._cell_pos_dict(dummy, m, n_jobs=0)
    out["odd_mask"] = m
    out["odd_ids"], out["odd_table"] = table_from_dict(d)
    out["odd_first_rows"] = np.array(d[77][0], np.int64)   # scan-order check for the two-piece cell
    out["odd_first_cols"] = np.array(d[77][1], np.int64)
    from PIL import Image
    ex = np.array(Image.open(os.path.join(REF, "examples/example_2_cell_mask.png"))).astype(np.int32)
    d2 = pre.ImageProcessor._cell_pos_dict(dummy, ex, n_jobs=0)
    out["example2_mask"] = ex.astype(np.uint16)
    out["example2_ids"], out["example2_table"] = table_from_dict(d2)
    ex1 = np.array(Image.open(os.path.join(REF, "examples/example_1_cell_mask.png"))).astype(np.int32)    # BASELINE config 1 mask (1850 cells)
    d1 = pre.ImageProcessor._cell_pos_dict(dummy, ex1, n_jobs=0)
    out["example1_mask"] = ex1.astype(np.uint16)
    out["example1_ids"], out["example1_table"] = table_from_dict(d1)
    ms, _ = synth.make_mask_and_image(160, 200, 60, 1, synth.SEED_BASE + 111, want_image=False)
    d3 = pre.ImageProcessor._cell_pos_dict(dummy, ms.numpy(), n_jobs=0)
    out["synth_ids"], out["synth_table"] = table_from_dict(d3)
    np.savez_compressed(os.path.join(HERE, "cellpos.npz"), **out)
    print("cellpos.npz", len(d), len(d2), len(d3))


# ---------------------------------------------------------------------------------------------- G3
def golden_patches():
    pre = ref("preprocess")
    utils = ref("utils")
    dummy = object.__new__(pre.ImageProcessor)
    dummy.scale = 1.0
    out = {}
    # case A: the hand-built mask with a 7-channel normalised image
    m = odd_mask()
    raw = small_raw_tile(7, 90, 120, synth.SEED_BASE + 121)
    img = pre.ImageProcessor._normalize(dummy, raw, blur=0.3, amax=99.8)
    d = pre.ImageProcessor._cell_pos_dict(dummy, m, n_jobs=0)
    tmp = tempfile.mkdtemp()
    cases = {
        "all7": [0, 1, 2, 3, 4, 5, 6],
        "perm": [4, 2, 6, 0, 1, 5, 3],
        "one_missing": [0, 1, -1, 3, 4, 5, 6],
        "two_missing": [0, -1, 2, 3, -1, 5, 6],      # second -1 aliases the last image channel
        "three": [6, 0, 2],
    }
    out["A_raw"] = raw
    out["A_image"] = img
    out["A_mask"] = m
    for name, idx in cases.items():
        inten = pre.ImageProcessor._img2patches(dummy, img, m, idx, d, None, id="g_" + name, save_path=tmp, save_tensor=True,
                                                int_full=True)
        out[f"A_{name}_index"] = np.array(idx, np.int64)
        out[f"A_{name}_patches"] = torch.load(os.path.join(tmp, f"g_{name}_batch_0.pt")).numpy()
        out[f"A_{name}_intensity"] = inten
    # soft mask alone for three cells (fp32)
    lab = np.zeros((40, 40))
    lab[:40, :40] = m[20:60, 30:70]
    out["A_smooth_big"] = utils.smooth(lab, 65000)
    out["A_smooth_nested"] = utils.smooth(lab, 41)
    # case B: un-normalised uint16 image (normalization=False path: integer arithmetic in _move_image_range)
    ms, raw2 = synth.make_mask_and_image(96, 96, 16, 3, synth.SEED_BASE + 122)
    ms = ms.numpy()
    raw2 = raw2.numpy().astype(np.uint16)
    d2 = pre.ImageProcessor._cell_pos_dict(dummy, ms, n_jobs=0)
    inten = pre.ImageProcessor._img2patches(dummy, raw2, ms, [2, 0, 1], d2, None, id="g_raw", save_path=tmp, save_tensor=True, int_full=True)
    out["B_raw"] = raw2
    out["B_mask"] = ms
    out["B_patches"] = torch.load(os.path.join(tmp, "g_raw_batch_0.pt")).numpy()
    out["B_intensity"] = inten
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(HERE, "patches.npz"), **out)
    print("patches.npz", {k: v.shape for k, v in out.items() if k.endswith("patches")})


def golden_patches_scaled():
    """cell_size != 30: the reference's own _img2patches with scale = cell_size / 30 (patch_size = int(40 * scale), crop_cell
    on that window, resize back to 40 x 40).  skimage is absent, so `resize` is oracle/skimage_like.py: this pins the
    reference's control flow around it (window size, what is resized, what the intensity is taken from), not skimage itself."""
    pre = ref("preprocess")
    dummy = object.__new__(pre.ImageProcessor)
    out = {}
    m = odd_mask()
    raw = small_raw_tile(3, 90, 120, synth.SEED_BASE + 141)
    dummy.scale = 1.0
    img = pre.ImageProcessor._normalize(dummy, raw, blur=0.3, amax=99.8)
    d = pre.ImageProcessor._cell_pos_dict(dummy, m, n_jobs=0)
    tmp = tempfile.mkdtemp()
    out["image"] = img
    out["mask"] = m
    for cell_size in (20, 34, 45, 60):
        dummy.scale = cell_size / 30.0
        inten = pre.ImageProcessor._img2patches(dummy, img, m, [2, 0, 1], d, None, id=f"g_s{cell_size}", save_path=tmp, save_tensor=True,
                                                int_full=True)
        out[f"s{cell_size}_patches"] = torch.load(os.path.join(tmp, f"g_s{cell_size}_batch_0.pt")).numpy()
        out[f"s{cell_size}_intensity"] = inten
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(HERE, "patches_scaled.npz"), **out)
    print("patches_scaled.npz", {k: v.shape for k, v in out.items() if k.endswith("patches")})


# ---------------------------------------------------------------------------------------------- G4
PARSER_CASES = {
    "basic7": ['CD45', 'CD20', 'CD4', 'CD8', 'DAPI', 'CD11c', 'CD3'],
    "extended10": ['DAPI', 'CD3', 'CD4', 'CD8', 'CD11c', 'CD20', 'CD45', 'CD68', 'CD163', 'CD56'],
    "full15": list(synth.FULL_PANEL_MARKERS),
    "full_1_missing": [m for m in synth.FULL_PANEL_MARKERS if m != 'Trypase'] + ['CollagenIV'],
    "full_2_missing": [m for m in synth.FULL_PANEL_MARKERS if m not in ('Trypase', 'CD15')],
    "full_3_missing": [m for m in synth.FULL_PANEL_MARKERS if m not in ('Trypase', 'CD15', 'FoxP3')],
    "full_4_missing": [m for m in synth.FULL_PANEL_MARKERS if m not in ('Trypase', 'CD15', 'FoxP3', 'CD138')],
    "examples_markers": None,  # the reference's examples/markers.txt
    "aliases": ['DNA', 'CD3e', 'CD4', 'CD8', 'CD11c', 'CD79', 'CD45', 'CK', 'SMActin', 'CD31', 'Vimentin', 'Ki67'],
    "alias_truncated": ['DNA', 'CD3', 'CD4', 'CD8', 'CK', 'aSMA', 'CD31', 'Ki67', 'CD45', 'CD20', 'Vim'],
    "alias_blocked": ['DNA', 'DAPI', 'CD45', 'GFAP'],
    "nerve": ['DAPI', 'CD45', 'GFAP', 'X'],
    "struct_one_missing": ['DAPI', 'aSMA', 'CD31', 'PanCK', 'Vimentin', 'CD45', 'Extra'],
    "nothing": ['A', 'B', 'C'],
    "e2e12": ['CD45', 'CD20', 'CD4', 'CD8', 'DAPI', 'CD11c', 'CD3', 'aSMA', 'CD31', 'PanCK', 'Vimentin', 'Ki67'],
}


def golden_parser():
    mp = ref("markerParse")
    res = {}
    tmp = tempfile.mkdtemp()
    for name, markers in PARSER_CASES.items():
        if markers is None:
            path = os.path.join(REF, "examples/markers.txt")
            with open(path) as f:
                markers = [ln.strip() for ln in f if ln.strip()]
        path = os.path.join(tmp, name + ".txt")
        with open(path, "w") as f:
            f.write("\n".join(markers) + "\n")
        for strict in (True, False):
            p = mp.MarkerParser(strict=strict, logger=_Log())
            p.parse(path)
            res[f"{name}|{'strict' if strict else 'loose'}"] = {
                "markers_in": markers,
                "markers": [str(m) for m in p.markers],
                "indices": {k: (None if v is None else [int(i) for i in v]) for k, v in p.indices.items()},
                "flags": [bool(p.immune_base), bool(p.immune_extended), bool(p.immune_full), bool(p.struct), bool(p.nerve)],
            }
    shutil.rmtree(tmp)
    with open(os.path.join(HERE, "parser_cases.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("parser_cases.json", len(res))


# ---------------------------------------------------------------------------------------------- G5
ALL_TYPES = ["B cell", "CD4 T cell", "CD8 T cell", "Dendritic cell", "Regulatory T cell", "Granulocyte cell", "Mast cell",
             "M1 macrophage cell", "M2 macrophage cell", "Natural killer cell", "Plasma cell", "Endothelial cell",
             "Epithelial cell", "Stroma cell", "Smooth muscle", "Proliferating/tumor cell", "Nerve cell", "Others"]
MODEL_CLASSES = {"immune_full": 12, "immune_extended": 8, "immune_base": 5, "struct": 6, "nerve": 2}


def seeded_probs(model, n, seed, sharp=3.0):
    k = MODEL_CLASSES[model]
    z = synth.approx_normal(synth.stream_key(seed, "vote/" + model), n * k).reshape(n, k).to(torch.float32) * sharp
    p = torch.softmax(z, dim=1).numpy()
    # force exact ties and near-threshold rows
    p[0, :] = np.float32(1.0 / k)
    if k >= 3:
        p[1, 0] = p[1, 1] = np.float32(0.4)
        p[1, 2:] = np.float32(0.2 / (k - 2))
    return p.astype(np.float32)


def golden_vote():
    model = ref("model")
    n = 64
    out = {}
    meta = {}
    type_conf_json = json.load(open(os.path.join(REF, "hyperparameters.json")))["cell_type_confidence"]
    per_type_b = dict(type_conf_json)
    per_type_b["CD4 T cell"] = 0.5
    per_type_b["Epithelial cell"] = 0.0
    per_type_b["Nerve cell"] = 0.9
    combos = [
        ("b2_full_struct", "immune_full", True, False),
        ("b2_ext_struct", "immune_extended", True, False),
        ("b2_base_struct_nerve", "immune_base", True, True),   # nerve ignored by branch 2
        ("b3_struct_nerve", None, True, True),
        ("b4_base_nerve", "immune_base", False, True),
        ("b4_full_nerve", "immune_full", False, True),
        ("b5_full", "immune_full", False, False),
        ("b5_ext", "immune_extended", False, False),
        ("b5_base", "immune_base", False, False),
        ("b6_struct", None, True, False),
        ("b7_nerve", None, False, True),
    ]
    settings = [("default", 0.25, None), ("conf03", 0.3, None), ("json", 0.3, type_conf_json), ("mixed", 0.3, per_type_b)]
    for ci, (cname, immune, use_s, use_n) in enumerate(combos):
        tables = {}
        if immune:
            tables[immune] = seeded_probs(immune, n, 1000 + ci)
        if use_s:
            tables["struct"] = seeded_probs("struct", n, 2000 + ci)
        if use_n:
            tables["nerve"] = seeded_probs("nerve", n, 3000 + ci)
        for k, v in tables.items():
            out[f"{cname}__p_{k}"] = v
        for sname, conf, tconf in settings:
            a = object.__new__(model.Annotator)
            a.immune_full_pred, a.struct_pred, a.nerve_pred = [], [], []
            a.immune_annotations, a.struct_annotations, a.nerve_annotations = [], [], []
            a.annotations, a.confidence = [], []
            a.confidence_thresh = conf
            a.extra_cell_types = False
            a.cell_type_confidence = tconf if tconf is not None else {k: -1 for k in ALL_TYPES}
            names = {"immune_full": ["CD4 T cell", "CD8 T cell", "Dendritic cell", "B cell", "M1 macrophage cell", "M2 macrophage cell",
                                     "Regulatory T cell", "Granulocyte cell", "Plasma cell", "Natural killer cell", "Mast cell", "Others"],
                     "immune_extended": ["CD4 T cell", "CD8 T cell", "Dendritic cell", "B cell", "M1 macrophage cell",
                                         "M2 macrophage cell", "Natural killer cell", "Others"],
                     "immune_base": ["B cell", "CD4 T cell", "CD8 T cell", "Others", "Dendritic cell"],
                     "struct": ["Stroma cell", "Smooth muscle", "Endothelial cell", "Epithelial cell", "Proliferating/tumor cell", "Others"],
                     "nerve": ["Nerve cell", "Others"]}

            def dicts(m):
                return [{names[m][i]: row[i] for i in range(len(row))} for row in tables[m]]
            if immune:
                a.immune_annotations.append(dicts(immune))
                if immune == "immune_full":
                    a.immune_full_pred.append(a.immune_annotations[0])
            if use_s:
                a.struct_annotations.append(dicts("struct"))
                a.struct_pred.append(a.struct_annotations[0])
            if use_n:
                a.nerve_annotations.append(dicts("nerve"))
                a.nerve_pred.append(a.nerve_annotations[0])
            a.merge_by_voting()
            labels = a.annotations[0]
            confs = a.confidence[0]
            key = f"{cname}__{sname}"
            meta[key] = {"immune": immune, "struct": use_s, "nerve": use_n, "conf": conf, "type_conf": tconf, "labels": labels,
                         "conf_is_int": [isinstance(c, int) for c in confs]}
            out[key + "__conf"] = np.array([np.float32(c) for c in confs], np.float32)
            # predict() tail + CSV through the reference's own methods
            a.cell_types = a._get_unique_cell_types()
            a.cell_types = np.delete(a.cell_types, np.where(a.cell_types == "Others"))
            a.cell_types = np.append(a.cell_types, "Others")
            meta[key]["cell_types"] = [str(s) for s in a.cell_types]
            if sname in ("default", "mixed"):
                pos = {}
                for j in range(n):
                    rows = [j, j + 1, j + 3, (7 * j) % 13]
                    cols = [2 * j, 2 * j + 1, 5, (11 * j) % 17]
                    pos[100 + 3 * j] = (rows, cols)
                a.preprocessor = types.SimpleNamespace(cell_pos_dict=[pos])
                tmp = tempfile.mkdtemp()
                a.result_dir = tmp
                a.batch_id = "g"
                a.logger = _Log()
                a.export_annotations()
                meta[key]["csv"] = open(os.path.join(tmp, "g_annotation_0.csv")).read()
                shutil.rmtree(tmp)
    # branch 1 raises KeyError('Others')
    a = object.__new__(model.Annotator)
    a.annotations, a.confidence = [], []
    a.confidence_thresh = 0.25
    a.extra_cell_types = False
    a.cell_type_confidence = {k: -1 for k in ["CD4 T cell", "Others"]}
    one = [{"CD4 T cell": np.float32(0.6), "Others": np.float32(0.4)}]
    a.immune_full_pred = [one]
    a.struct_pred = [[{"Stroma cell": np.float32(0.5), "Others": np.float32(0.5)}]]
    a.nerve_pred = [[{"Nerve cell": np.float32(0.5), "Others": np.float32(0.5)}]]
    try:
        a.merge_by_voting()
        meta["branch1"] = "no error"
    except KeyError as e:
        meta["branch1"] = "KeyError:" + str(e.args[0])
    np.savez_compressed(os.path.join(HERE, "vote_cases.npz"), **out)
    with open(os.path.join(HERE, "vote_cases.json"), "w") as f:
        json.dump(meta, f, indent=0, sort_keys=True)
    print("vote_cases", len(meta), meta["branch1"])


# ---------------------------------------------------------------------------------------------- G7
def vit_inputs(model_name, n, seed):
    d, c, k = synth.VIT_CONFIGS[model_name]
    u = synth.uniform(synth.stream_key(seed, "vitx/" + model_name), n * c * 1600).reshape(n, c, 40, 40).to(torch.float32)
    x = u * 2 - 1
    return torch.where(x > 0.1, x, torch.full_like(x, -1.0))


def build_ref_model(model_mod, name):
    d, c, k = synth.VIT_CONFIGS[name]
    fac = {"nerve": model_mod.vit_tiny, "immune_base": model_mod.vit_s, "struct": model_mod.vit_s,
           "immune_extended": model_mod.vit_m, "immune_full": model_mod.vit_l}[name]
    return fac(img_size=40, in_chans=c, num_classes=k, drop_path_rate=0.1, global_pool=False)


def golden_vit():
    model = ref("model")
    out = {}
    for name in synth.VIT_CONFIGS:
        sd = synth.make_vit_state_dict(name, synth.SEED_BASE + 7)
        m = build_ref_model(model, name)
        m.load_state_dict(sd)
        m.eval()
        x = vit_inputs(name, 8, synth.SEED_BASE + 7)
        with torch.no_grad():
            lg = m(x)
            pr = torch.nn.functional.softmax(lg, dim=1)
        out[name + "_logits"] = lg.numpy()
        out[name + "_probs"] = pr.numpy()
        out[name + "_x_sha"] = np.frombuffer(bytes.fromhex(sha(x.numpy())), np.uint8)
    np.savez_compressed(os.path.join(HERE, "vit_logits.npz"), **out)
    print("vit_logits.npz ok")


# ---------------------------------------------------------------------------------------------- G8
def golden_e2e():
    """Whole reference Annotator (preprocess -> predict -> export_annotations) on small synthetic tiles with seeded weights."""
    model = ref("model")
    cases = {
        # name: (markers, H, W, cells, seed, strict, blur, amax, conf, models needed)
        "two_model": (PARSER_CASES["e2e12"], 176, 208, 56, synth.SEED_BASE + 201, False, 0.3, 99.8, 0.3, ["immune_base", "struct"]),
        "basic": (PARSER_CASES["basic7"], 160, 160, 40, synth.SEED_BASE + 202, True, 0.3, 99.8, 0.3, ["immune_base"]),
    }
    meta = {}
    out = {}
    cwd = os.getcwd()
    for cname, (markers, h, w, cells, seed, strict, blur, amax, conf, models) in cases.items():
        tmp = tempfile.mkdtemp()
        os.chdir(tmp)
        mdir = "src/multiplexed_image_annotator/cell_type_annotation/models"
        os.makedirs(mdir)
        for m in models:
            torch.save({"model": synth.make_vit_state_dict(m, seed)}, os.path.join(mdir, m + ".pth"))
        mask, img = synth.make_mask_and_image(h, w, cells, len(markers), seed)
        np.save("img.npy", img.numpy().astype(np.uint16))
        np.save("mask.npy", mask.numpy().astype(np.int32))
        with open("markers.txt", "w") as f:
            f.write("\n".join(markers) + "\n")
        with open("images.csv", "w") as f:
            f.write("image_path,mask_path\nimg.npy,mask.npy\n")
        a = model.Annotator("markers.txt", "images.csv", "cpu", "./", "g", strict, False, -1, True, blur, amax, conf, 30, None, n_jobs=0)
        a.preprocess()
        # centre each model's logits on this tile (stored with the fixture) so labels spread over classes
        tensor_name = {"immune_base": "immune_base", "struct": "structure"}
        for m in models:
            sd = synth.make_vit_state_dict(m, seed)
            net = build_ref_model(model, m)
            net.load_state_dict(sd)
            net.eval()
            x = torch.load(os.path.join("tmp", f"g_0_{tensor_name[m]}_batch_0.pt"))
            with torch.no_grad():
                feats = net.forward_features(x)
            sd["head.bias"] = synth.calibrate_head_bias(sd, feats)
            out[cname + "__head_bias_" + m] = sd["head.bias"].numpy()
            torch.save({"model": sd}, os.path.join(mdir, m + ".pth"))
        a.predict(8)
        a.export_annotations()
        a.logger.close()
        meta[cname] = {"markers": markers, "h": h, "w": w, "cells": cells, "seed": seed, "strict": strict, "blur": blur, "amax": amax,
                       "conf": conf, "models": models, "labels": list(a.annotations[0]), "cell_types": [str(s) for s in a.cell_types],
                       "csv": open("results/g_annotation_0.csv").read(),
                       "type_ints": [int(r["Cell type"]) for r in a.annotations_all[0]],
                       "cell_ids": [int(r["Cell ID"]) for r in a.annotations_all[0]],
                       "img_sha": sha(img.numpy().astype(np.uint16)), "mask_sha": sha(mask.numpy().astype(np.int32))}
        out[cname + "__conf"] = np.array([np.float32(c) for c in a.confidence[0]], np.float32)
        out[cname + "__intensity"] = a.preprocessor.intensity_full[0]
        names = {"immune_base": ["B cell", "CD4 T cell", "CD8 T cell", "Others", "Dendritic cell"],
                 "struct": ["Stroma cell", "Smooth muscle", "Endothelial cell", "Epithelial cell", "Proliferating/tumor cell", "Others"]}
        if "immune_base" in models:
            out[cname + "__p_immune_base"] = np.array([[d[k] for k in names["immune_base"]] for d in a.immune_base_pred[0]], np.float32)
        if "struct" in models:
            out[cname + "__p_struct"] = np.array([[d[k] for k in names["struct"]] for d in a.struct_pred[0]], np.float32)
        os.chdir(cwd)
        shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **out)
    with open(os.path.join(HERE, "e2e.json"), "w") as f:
        json.dump(meta, f, indent=0, sort_keys=True)
    print("e2e", {k: len(v["labels"]) for k, v in meta.items()})


# ---------------------------------------------------------------------------------------------- BASELINE config 1 stand-in
def golden_config1():
    """BASELINE.json configs[0] stand-in (examples/example_1.tif is a missing blob): the reference's own
    ``Annotator`` on ``examples/example_1_cell_mask.png`` (600 x 600, 1850 cells) + a synthetic 7-channel image in Basic-panel
    marker order, strict, blur 0.3, amax 99.8, confidence 0.3, ``predict(8)``, device cpu -- reference main.py:9-36."""
    from PIL import Image
    model = ref("model")
    seed = synth.SEED_BASE + 1            # config k uses seed base + k
    markers = synth.BASIC_PANEL_MARKERS
    mask = np.array(Image.open(os.path.join(REF, "examples/example_1_cell_mask.png"))).astype(np.int32)
    img = synth.make_image_for_mask(torch.from_numpy(mask), len(markers), seed).numpy().astype(np.uint16)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        mdir = "src/multiplexed_image_annotator/cell_type_annotation/models"
        os.makedirs(mdir)
        sd = synth.make_vit_state_dict("immune_base", seed)
        torch.save({"model": sd}, os.path.join(mdir, "immune_base.pth"))
        np.save("img.npy", img)
        np.save("mask.npy", mask)
        with open("markers.txt", "w") as f:
            f.write("\n".join(markers) + "\n")
        with open("images.csv", "w") as f:
            f.write("image_path,mask_path\nimg.npy,mask.npy\n")
        a = model.Annotator("markers.txt", "images.csv", "cpu", "./", "c1", True, False, -1, True, 0.3, 99.8, 0.3, 30, None, n_jobs=0)
        a.preprocess()
        net = build_ref_model(model, "immune_base")
        net.load_state_dict(sd)
        net.eval()
        x = torch.load(os.path.join("tmp", "c1_0_immune_base_batch_0.pt"))
        with torch.no_grad():
            feats = torch.cat([net.forward_features(x[i:i + 128]) for i in range(0, x.shape[0], 128)])
        sd["head.bias"] = synth.calibrate_head_bias(sd, feats)
        torch.save({"model": sd}, os.path.join(mdir, "immune_base.pth"))
        a.predict(8)
        a.export_annotations()
        a.logger.close()
        names = ["B cell", "CD4 T cell", "CD8 T cell", "Others", "Dendritic cell"]
        probs = np.array([[d[k] for k in names] for d in a.immune_base_pred[0]], np.float32)
        csv = open("results/c1_annotation_0.csv").read()
        meta = {"markers": markers, "seed": seed, "cells": len(a.annotations[0]), "labels": list(a.annotations[0]),
                "cell_types": [str(s) for s in a.cell_types], "csv": csv, "img_sha": sha(img), "mask_sha": sha(mask),
                "strict": True, "blur": 0.3, "amax": 99.8, "conf": 0.3, "batch_size": 8}
        np.savez_compressed(os.path.join(HERE, "config1.npz"), head_bias=sd["head.bias"].numpy(), probs=probs,
                            conf=np.array([np.float32(c) for c in a.confidence[0]], np.float32), intensity=a.preprocessor.intensity_full[0])
        with open(os.path.join(HERE, "config1.json"), "w") as f:
            json.dump(meta, f, indent=0, sort_keys=True)
        srt = np.sort(probs, axis=1)
        print("config1", meta["cells"], "cells;", {t: meta["labels"].count(t) for t in set(meta["labels"])},
              "min top-2 margin %.2e" % float((srt[:, -1] - srt[:, -2]).min()))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)


# ---------------------------------------------------------------------------------------------- colorize
def golden_colorize():
    """Annotator.colorize (model.py:806-858) and utils.get_colors / number_to_rgb, driven with the labels / confidences of the
    e2e goldens (no ViT run needed): the three label paintings of each tile + the palette for several sizes."""
    model = ref("model")
    utils = ref("utils")
    pre = ref("preprocess")
    meta = json.load(open(os.path.join(HERE, "e2e.json")))
    arrs = np.load(os.path.join(HERE, "e2e.npz"))
    out = {}
    cwd = os.getcwd()
    for cname, m in meta.items():
        tmp = tempfile.mkdtemp()
        os.chdir(tmp)
        os.makedirs("results")
        os.makedirs("src/multiplexed_image_annotator/cell_type_annotation/_working_dir_temp")
        mask, _ = synth.make_mask_and_image(m["h"], m["w"], m["cells"], len(m["markers"]), m["seed"], want_image=False)
        mask = mask.numpy().astype(np.int32)
        a = object.__new__(model.Annotator)
        dummy = object.__new__(pre.ImageProcessor)
        dummy.masks = [mask]
        dummy.cell_pos_dict = [pre.ImageProcessor._cell_pos_dict(dummy, mask, n_jobs=0)]
        a.preprocessor = dummy
        a.annotations = [list(m["labels"])]
        conf = arrs[cname + "__conf"]
        a.confidence = [[-1 if c == -1 else np.float32(c) for c in conf]]
        a.cell_types = np.array(m["cell_types"])
        a.colors = utils.get_colors(len(a.cell_types))
        a.n_regions = 0
        a.result_dir = "results"
        a.batch_id = "g"
        a.colorize(from_script=False)
        from PIL import Image
        out[cname + "__type_rgb"] = np.array(Image.open("results/g_colorized_annotation_0.png"))
        out[cname + "__conf_rgb"] = np.array(Image.open("results/g_confidence_0.png"))
        out[cname + "__type_idx"] = np.array(Image.open("src/multiplexed_image_annotator/cell_type_annotation/_working_dir_temp/output_img.png"))
        os.chdir(cwd)
        shutil.rmtree(tmp)
    for n in (1, 2, 6, 17, 18, 19, 30):
        out[f"colors_{n}"] = np.array(utils.get_colors(n), np.int64)
    vals = np.concatenate([np.linspace(0, 1, 257), np.float32([0.25, 0.3, 0.5360000133514404, 0.999999])]).astype(np.float32)
    out["viridis_in"] = vals
    out["viridis_rgb"] = np.array([utils.number_to_rgb(v) for v in vals], np.int64)
    np.savez_compressed(os.path.join(HERE, "colorize.npz"), **out)
    print("colorize.npz", {k: v.shape for k, v in out.items()})


# ---------------------------------------------------------------------------------------------- kNN neighbourhood
def golden_neighborhood():
    """spatial_methods.neighborhood_analysis (spatial_methods.py:13-130) on annotations_all rebuilt from the e2e goldens + one larger
    synthetic tile with hashed cell types: the CSV text for per-image and integrated modes."""
    sp = ref("spatial_methods")
    pre = ref("preprocess")
    meta = json.load(open(os.path.join(HERE, "e2e.json")))
    dummy = object.__new__(pre.ImageProcessor)
    ann_all, names = [], None
    cases = {}
    for cname in ("basic", "two_model"):
        m = meta[cname]
        mask, _ = synth.make_mask_and_image(m["h"], m["w"], m["cells"], len(m["markers"]), m["seed"], want_image=False)
        d = pre.ImageProcessor._cell_pos_dict(dummy, mask.numpy().astype(np.int32), n_jobs=0)
        cases[cname] = ([{"Cell ID": k, "Cell type": t, "Confidence": 0.5, "Row": d[k][0], "Column": d[k][1]} for k, t in zip(d.keys(), m["type_ints"])],
                        m["cell_types"])
    # a larger tile: 1500 cells, 6 types assigned by a hash of the label
    mask, _ = synth.make_mask_and_image(640, 700, 1500, 1, synth.SEED_BASE + 151, want_image=False)
    d = pre.ImageProcessor._cell_pos_dict(dummy, mask.numpy().astype(np.int32), n_jobs=0)
    keys = list(d.keys())
    types = (synth.hash_u24(synth.stream_key(synth.SEED_BASE + 151, "nbr"), torch.tensor(keys, dtype=torch.int64)) % 6).tolist()
    cases["big"] = ([{"Cell ID": k, "Cell type": int(t), "Confidence": 0.5, "Row": d[k][0], "Column": d[k][1]} for k, t in zip(keys, types)],
                    ["A", "B", "C", "D", "E", "Others"])
    out = {"big_types": types, "big_ids": [int(k) for k in keys]}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    for cname, (ann, ctypes) in cases.items():
        for nn in (10, 25):
            sp.neighborhood_analysis([ann], n_neighbors=nn, cell_types=np.array(ctypes), integrate=False, normalize=True, batch_id=f"{cname}{nn}",
                                     result_dir=".")
            out[f"{cname}__k{nn}"] = open(f"{cname}{nn}_neighborhood_0.csv").read()
    ann2 = [cases["big"][0], cases["big"][0][:700]]
    sp.neighborhood_analysis(ann2, n_neighbors=25, cell_types=np.array(cases["big"][1]), integrate=True, normalize=True, batch_id="int", result_dir=".")
    out["big_integrated_k25"] = open("int_integrated_neighborhood.csv").read()
    sp.neighborhood_analysis([cases["big"][0]], n_neighbors=10, cell_types=np.array(cases["big"][1]), integrate=False, normalize=False, batch_id="raw",
                             result_dir=".")
    out["big_raw_k10"] = open("raw_neighborhood_0.csv").read()
    os.chdir(cwd)
    shutil.rmtree(tmp)
    with open(os.path.join(HERE, "neighborhood.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("neighborhood.json", {k: len(v) for k, v in out.items()})


def golden_tissue():
    """spatial_methods.tissue_region_partition (spatial_methods.py:133-198): the `compositions` matrix the reference builds from its
    201-nearest-neighbour query (captured at the PCA call; PCA / KMeans themselves are scikit-learn with a random start and are not
    pinned), on the 1484-cell tile of the neighbourhood goldens."""
    sp = ref("spatial_methods")
    pre = ref("preprocess")
    dummy = object.__new__(pre.ImageProcessor)
    mask, _ = synth.make_mask_and_image(640, 700, 1500, 1, synth.SEED_BASE + 151, want_image=False)
    d = pre.ImageProcessor._cell_pos_dict(dummy, mask.numpy().astype(np.int32), n_jobs=0)
    keys = list(d.keys())
    types = (synth.hash_u24(synth.stream_key(synth.SEED_BASE + 151, "nbr"), torch.tensor(keys, dtype=torch.int64)) % 6).tolist()
    ann = [{"Cell ID": k, "Cell type": int(t), "Confidence": 0.5, "Row": d[k][0], "Column": d[k][1]} for k, t in zip(keys, types)]
    captured = {}

    class CapturePCA:
        def __init__(self, n_components=None):
            pass

        def fit_transform(self, x):
            captured["x"] = np.array(x, copy=True)
            return x[:, :4]

    real = sp.PCA
    sp.PCA = CapturePCA
    try:
        labels = sp.tissue_region_partition([ann], n_clusters=3, n_jobs=0, method="kmeans")
    finally:
        sp.PCA = real
    assert len(labels[0]) == len(keys)
    np.savez_compressed(os.path.join(HERE, "tissue.npz"), compositions=captured["x"].astype(np.float64), types=np.array(types, np.int64))
    print("tissue.npz", captured["x"].shape)


# ---------------------------------------------------------------------------------------------- G6
def mae_inputs(panel, n, seed):
    L = synth.MAE_PANELS[panel]
    u = synth.uniform(synth.stream_key(seed, "maex/" + panel), n * L * 1600).reshape(n, L, 40, 40).to(torch.float32)
    x = u * 2 - 1
    return torch.where(x > 0.0, x, torch.full_like(x, -1.0))


MAE_CASES = {"immune_base": [0, 1, 3, 4, 5, 6], "immune_full": [0, 1, 2, 3, 4, 6, 7, 8, 9, 11, 12, 13]}   # present positions


def golden_mae():
    """Reference MarkerImputer.impute (markerImputer.py:258-329) with seeded full-depth weights, batch size 4 on 6 cells
    (two full batches would hide the ragged-tail + empty-last-batch loop of the reference)."""
    imp = ref("markerImputer")
    out = {}
    cwd = os.getcwd()
    for panel, present in MAE_CASES.items():
        seed = synth.SEED_BASE + 301
        tmp = tempfile.mkdtemp()
        os.chdir(tmp)
        mdir = "src/multiplexed_image_annotator/cell_type_annotation/models"
        os.makedirs(mdir)
        torch.save({"model": synth.make_mae_state_dict(panel, seed)}, os.path.join(mdir, panel + "_impute.pth"))
        m = imp.MarkerImputer(present, "cpu", panel)
        x = mae_inputs(panel, 6, seed)
        L = x.shape[1]
        missing = [c for c in range(L) if c not in present]
        x[:, missing] = -1.0
        y = m.impute(x.clone(), 4)
        assert torch.equal(y[:, present], x[:, present])
        out[panel + "_present"] = np.array(present, np.int64)
        out[panel + "_pred"] = y[:, missing].numpy()
        os.chdir(cwd)
        shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(HERE, "mae.npz"), **out)
    print("mae.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    install_shims()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["normalize", "cellpos", "patches", "patches_scaled", "parser", "vote", "vit", "e2e", "mae", "colorize", "neighborhood", "tissue", "config1"]
    for w in which:
        globals()["golden_" + w]()
