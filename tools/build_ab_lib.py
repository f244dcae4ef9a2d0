#!/usr/bin/env python3
"""A second build of BOTH libraries from another git revision of csrc/ + include/, for same-box A/B runs:

    python tools/build_ab_lib.py old HEAD~1 [-DFLAG ...]   ->  multiplexed-image-annotator_amd/libribca_ab_old.so + libribca_ab_old_test.so
    RIBCA_LIB=libribca_ab_old.so python tools/bench_mx_only.py          (the hooks resolve to the matching _test.so: _lib.TEST_LIB_PATH)

`rev` = WORK takes the working tree (with extra -D flags: a timing variant of the current sources).  The libraries are git-ignored and travel
to the GPU box (named libribca_ab_*: .gpurunignore only drops libribca_hip_diag*); delete them when the A/B is done.
RIBCA_AB_PATCH=<file>: a patch (-p1, paths from the repo root) applied to the temporary copy before compiling -- experiments that never touch the tree."""
import concurrent.futures
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiplexed_image_annotator_amd import build as B

name, rev, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
tmp = tempfile.mkdtemp(prefix="ribca_ab_")
try:
    if rev == "WORK":
        shutil.copytree(os.path.join(B.HERE, "csrc"), os.path.join(tmp, "multiplexed-image-annotator_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    else:
        ar = subprocess.run(["git", "-C", ROOT, "archive", rev, "multiplexed-image-annotator_amd/csrc", "include"], check=True, capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", tmp], input=ar, check=True)
    if os.environ.get("RIBCA_AB_PATCH"):      # a timing-only or experimental change that never touches the tree: applied to the temporary copy
        subprocess.run(["patch", "-p1", "-d", tmp, "-i", os.path.abspath(os.environ["RIBCA_AB_PATCH"])], check=True)
    csrc = os.path.join(tmp, "multiplexed-image-annotator_amd", "csrc")
    srcs = [s for s in B.SOURCES + B.TEST_SOURCES if os.path.exists(os.path.join(csrc, s))]
    # revisions older than round 6 export their C++ launchers to the hook library and carry no visibility pragmas in their headers: default
    # visibility, no version script, no --no-undefined for those
    hidden = os.path.exists(os.path.join(csrc, "ribca_internal.h"))
    flags = B.FLAGS if hidden else [f for f in B.FLAGS if not f.startswith("-fvisibility")]

    def cc(src):
        obj = os.path.join(tmp, src.replace(".hip", ".o"))
        subprocess.run([B._hipcc()] + flags + extra + ["-c", os.path.join(csrc, src), "-o", obj], check=True, capture_output=True)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = dict(zip(srcs, ex.map(cc, srcs)))
    lib = os.path.join(B.HERE, f"libribca_ab_{name}.so")
    vs = ["-Wl,--version-script=" + B.EXPORTS] if hidden else []
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-soname," + os.path.basename(lib)] + vs + ["-o", lib]
                   + [objs[s] for s in srcs if s not in B.TEST_SOURCES], check=True)
    tests = [objs[s] for s in srcs if s in B.TEST_SOURCES]
    if tests:
        subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + (["-Wl,--no-undefined"] + vs if hidden else []) + ["-o", lib[:-3] + "_test.so"] + tests
                       + ["-L" + B.HERE, f"-lribca_ab_{name}", "-Wl,-rpath,$ORIGIN"], check=True)
    print("built", lib)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
