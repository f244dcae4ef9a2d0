"""Builds libribca_hip.so (hipcc, gfx950 only) in-tree next to this file.

``python -m multiplexed_image_annotator_amd.build`` or ``__graft_entry__.build()``.  The shared object has no
torch / Python dependency: plain ``hipcc -shared -fPIC``; it travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libribca_hip.so")
LIB_DIAG = os.path.join(HERE, "libribca_hip_diag.so")
# kernel-level hooks of tests/ and tools/ (include/ribca_hip_test.h): a library of their own, linked against the product library -- the same
# launchers and kernels, none of them exported by libribca_hip.so itself
LIB_TEST = os.path.join(HERE, "libribca_hip_test.so")
LIB_TEST_DIAG = os.path.join(HERE, "libribca_hip_diag_test.so")
TEST_SOURCES = ["ribca_test_api.hip"]
EXPORTS = os.path.join(CSRC, "exports.map")
SOURCES = ["gemm_split16.hip", "gemm_duo.hip", "gemm_mx.hip", "attention.hip", "cell_attention.hip", "vit_misc.hip", "preprocess.hip", "preprocess_scaled.hip", "vote.hip", "colorize.hip", "knn.hip", "normalize.hip", "ribca_api.hip"]
HEADERS = ["ribca_common.h", "ribca_kernels.h", "ribca_status.h", "ribca_internal.h", "gemm_epi.h", os.path.join("..", "..", "include", "ribca_hip.h"),
           os.path.join("..", "..", "include", "ribca_hip_test.h")]
# -fvisibility=hidden: the dynamic symbol table of either library is what its header declares between `#pragma GCC visibility push(default)` and
# `pop` -- no C++ launcher, kernel stub or template instantiation is exported (tests/test_abi.py compares the full `nm -D` list with the header)
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-fvisibility-inlines-hidden",
         # the scalar fp32 epilogue arithmetic stays scalar: packed fp32 operations beside another workgroup's MFMA stream cost more than the two
         # plain ones they replace (ribca_common.h gelu_erf1; same-box A/B profiles/r6/ab_gelu_scalar.txt)
         "-fno-slp-vectorize"]


def source_fingerprint() -> str:
    """sha256 over the kernel sources, headers and compile flags: identifies the code a measurement was taken on.  (The .so itself is
    not byte-reproducible across checkout directories -- hipcc derives its compilation-unit ids from the source path -- so the
    committed counter files are stamped with this instead; bench.py compares it with the tree it runs from.)"""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + TEST_SOURCES) + sorted(HEADERS):
        path = os.path.normpath(os.path.join(CSRC, name))
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, diag: bool = False) -> str:
    """diag=True builds libribca_hip_diag.so with -DRIBCA_DIAG: the product kernels plus the A/B / timing-ablation / stamp forms that
    tools/ drive (select it at run time with RIBCA_DIAG=1).  The product library carries none of them."""
    srcs = [s for s in SOURCES + TEST_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build_diag" if diag else "build")
    lib_path = LIB_DIAG if diag else LIB
    flags = FLAGS + (["-DRIBCA_DIAG"] if diag else [])
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + hdrs):
            cmd = [hipcc] + flags + ["-c", path, "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
            if verbose and r.stderr.strip():
                print(r.stderr, file=sys.stderr)
            return obj, True
        return obj, False

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(compile_one, srcs))
    objs = [o for (o, _), src in zip(res, srcs) if src not in TEST_SOURCES]
    test_objs = [o for (o, _), src in zip(res, srcs) if src in TEST_SOURCES]
    if force or _stale(lib_path, objs + [EXPORTS]):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-soname," + os.path.basename(lib_path), "-Wl,--version-script=" + EXPORTS, "-o", lib_path] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    test_path = LIB_TEST_DIAG if diag else LIB_TEST
    if force or _stale(test_path, test_objs + [lib_path, EXPORTS]):
        libname = os.path.basename(lib_path)[3:-3]
        # --no-undefined: the hooks reach the product library through its C entry points (ribca_internal_table) only; a stray direct reference to a
        # hidden launcher fails HERE, not at dlopen on the GPU box
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-Wl,--version-script=" + EXPORTS, "-o", test_path] + test_objs + ["-L" + HERE, "-l" + libname, "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", lib_path)
    return lib_path


if __name__ == "__main__":
    build(force="--force" in sys.argv, diag="--diag" in sys.argv)
