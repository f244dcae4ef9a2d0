#!/usr/bin/env python3
"""A second build of the library with extra compile flags, for a same-box A/B of two builds (RIBCA_LIB=<name> selects it):
    python tools/build_variant.py libribca_hip_b.so -DRIBCA_DUO_PREFETCH
Objects go to multiplexed-image-annotator_amd/build_<name>/ (git-ignored like every built file)."""
import concurrent.futures
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiplexed_image_annotator_amd import build as B

name, extra = sys.argv[1], sys.argv[2:]
objdir = os.path.join(B.HERE, "build_" + os.path.splitext(name)[0])
os.makedirs(objdir, exist_ok=True)
hipcc = B._hipcc()


def one(src):
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    r = subprocess.run([hipcc] + B.FLAGS + extra + ["-c", os.path.join(B.CSRC, src), "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return obj


with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
    objs = list(ex.map(one, B.SOURCES))
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + B.EXPORTS, "-o", os.path.join(B.HERE, name)] + objs, check=True)
print("built", os.path.join(B.HERE, name))
