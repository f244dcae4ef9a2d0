"""ctypes binding of ``libribca_hip.so`` (C ABI in ``include/ribca_hip.h``).

The product path has no CPU fallback: if the shared library is missing or a call fails this raises.
torch is used only to own device memory and streams; raw pointers cross the boundary.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_uint32, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# RIBCA_DIAG=1 selects the diagnostic library (same ABI + the A/B and timing-ablation kernel forms; `build --diag`): tools/ only
LIB_PATH = os.path.join(HERE, "libribca_hip_diag.so" if os.environ.get("RIBCA_DIAG") == "1" else "libribca_hip.so")
# RIBCA_LIB=<file name next to this module, or a path>: another build of the same library (same-box A/B of two builds, tools/ab_env.sh)
if os.environ.get("RIBCA_LIB"):
    LIB_PATH = os.environ["RIBCA_LIB"] if os.path.isabs(os.environ["RIBCA_LIB"]) else os.path.join(HERE, os.environ["RIBCA_LIB"])

TEST_LIB_PATH = LIB_PATH[:-3] + "_test.so"

_lib = None
_test_lib = None

#: name -> (restype, argtypes); must list every function declared in include/ribca_hip.h (the product ABI)
SIGNATURES = {
    "ribca_version": (c_int32, []),
    "ribca_last_error": (c_char_p, []),
    "ribca_mask_minmax": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p]),
    "ribca_label_table": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "ribca_channel_min": (c_int32, [c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "ribca_extract_patches": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                        c_void_p, c_void_p, c_void_p]),
    "ribca_extract_patches_scaled": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                               c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ribca_colorize": (c_int32, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ribca_knn_cooccurrence": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "ribca_knn_compositions": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "ribca_u16_to_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "ribca_gauss1d": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "ribca_bg_subtract": (c_int32, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "ribca_plane_max": (c_int32, [c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "ribca_radix_hist": (c_int32, [c_void_p, c_int32, c_int64, c_void_p, c_uint32, c_int32, c_int32, c_void_p, c_void_p]),
    "ribca_norm_finalize": (c_int32, [c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ribca_vit_blob_len": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "ribca_vit_create": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, POINTER(c_void_p)]),
    "ribca_vit_destroy": (None, [c_void_p]),
    "ribca_vit_workspace_bytes": (c_int64, [c_void_p, c_int32]),
    "ribca_vit_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "ribca_vit_forward_precise": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "ribca_vit_flops_per_cell": (c_double, [c_void_p]),
    "ribca_mae_blob_len": (c_int64, [c_int32, c_int32, c_int32]),
    "ribca_mae_create": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, POINTER(c_void_p)]),
    "ribca_mae_destroy": (None, [c_void_p]),
    "ribca_mae_workspace_bytes": (c_int64, [c_void_p, c_int32, c_int32]),
    "ribca_mae_impute": (c_int32, [c_void_p, c_void_p, POINTER(c_int32), c_int32, c_int32, c_void_p, c_int64, c_int32, c_void_p]),
    "ribca_vote": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_float, c_int32, c_void_p, c_void_p,
                             c_void_p]),
    "ribca_mx_enabled": (c_int32, [c_int32]),
    "ribca_mxz_enabled": (c_int32, [c_int32]),
    "ribca_prof_enable": (c_int32, [c_int32]),
    "ribca_prof_read": (c_int32, [POINTER(c_double), POINTER(c_int64)]),
    "ribca_prof_name": (c_char_p, [c_int32]),
    # not part of the stable ABI: the launcher table libribca_hip_test.so binds (csrc/ribca_internal.h); the package never calls it
    "ribca_internal_table": (c_void_p, [c_int32]),
}

#: the kernel-level hooks of include/ribca_hip_test.h: libribca_hip_test.so, loaded on first use by tests/ and tools/ only -- the package
#: itself never names one of them (tests/test_abi.py checks that)
TEST_SIGNATURES = {
    "ribca_test_pack_weight": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "ribca_test_layernorm": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "ribca_test_gemm": (c_int32, [c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                  c_void_p]),
    "ribca_test_qkv_attention": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_int32, c_void_p]),
    "ribca_test_fold_weight": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p,
                                         c_void_p, c_void_p]),
    "ribca_test_row_stats": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "ribca_test_resid_tiles": (c_int32, [c_int32]),
    "ribca_test_gemm_resid_ps": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                           c_void_p, c_void_p, c_void_p, c_void_p]),
    "ribca_test_resid_part_rows": (c_int32, [c_int32]),
    "ribca_test_gemm_resid_ps_duo": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                               c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ribca_test_mx_weight_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "ribca_test_mx_pack_act": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ribca_test_gemm_mx_resid": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ribca_test_gemm_mx_resid_packed": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "ribca_test_gemm_gelu_mx": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    "ribca_test_gemm_mx_fc1": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ribca_test_qkv_attention_mx": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "ribca_test_gemm_resid_zmx": (c_int32, [c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_int32, c_void_p]),
    "ribca_test_gemm_fold": (c_int32, [c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int32, c_void_p]),
    "ribca_test_qkv_attention_fold": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "ribca_test_gemm_duo_gelu": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_int32, c_void_p]),
    "ribca_test_cell_attention": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                            c_void_p]),
    "ribca_gemm_padded_n": (c_int32, [c_int32]),
    "ribca_set_gemm_variant": (c_int32, [c_int32]),
    "ribca_set_gemm_stamps": (c_int32, [c_void_p, c_int64]),
    "ribca_is_diag_build": (c_int32, []),
}


class RibcaError(RuntimeError):
    pass


class _Library:
    """The product library; a name of TEST_SIGNATURES resolves (and loads, once) the separate test-hook library instead."""

    def __init__(self, handle):
        self._handle = handle

    def __getattr__(self, name):
        if name in SIGNATURES:
            fn = getattr(self._handle, name)
        elif name in TEST_SIGNATURES:
            fn = getattr(_load_test_lib(), name)
        else:
            raise AttributeError(f"{name} is declared neither in include/ribca_hip.h nor in include/ribca_hip_test.h")
        setattr(self, name, fn)
        return fn


def _bind(handle, table):
    for name, (res, args) in table.items():
        if name == "ribca_internal_table" and os.environ.get("RIBCA_LIB") and not hasattr(handle, name):
            continue      # an A/B build of a revision older than the table (tools/build_ab_lib.py old <rev>): the package itself never calls it
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args


def _load_test_lib():
    global _test_lib
    if _test_lib is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise RibcaError(f"{TEST_LIB_PATH} not found: build it with `python -m multiplexed_image_annotator_amd.build`")
        lib()      # the product library first: the hooks call its launchers
        handle = ctypes.CDLL(TEST_LIB_PATH)
        _bind(handle, TEST_SIGNATURES)
        _test_lib = handle
        if os.environ.get("RIBCA_GEMM_VARIANT"):
            handle.ribca_set_gemm_variant(int(os.environ["RIBCA_GEMM_VARIANT"]))
    return _test_lib


def lib() -> _Library:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RibcaError(
                f"{LIB_PATH} not found: build it with `python -m multiplexed_image_annotator_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback for the hot path.")
        # torch must be imported first: it bundles its own HIP runtime (same SONAME); loading ours first would pull in a
        # second runtime from /opt/rocm and the two do not share device state
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        _bind(handle, SIGNATURES)
        _lib = _Library(handle)
        if os.environ.get("RIBCA_GEMM_VARIANT"):
            _load_test_lib()
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().ribca_last_error()
        raise RibcaError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (must be contiguous) or None."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor passed to the HIP library must be contiguous")
    return t.data_ptr()


def stream_ptr():
    """Current torch HIP stream as an integer handle (the library enqueues on it)."""
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RibcaError("no HIP device visible: the RIBCA hot path runs only on an MI355X (gfx950); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())
