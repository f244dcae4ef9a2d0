"""The C-ABI library loads without a GPU and exports every symbol include/ribca_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="ribca_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ribca_[a-z0-9_]+)\s*\(", text)))


def exported_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    # the FULL dynamic symbol table: every defined symbol of any type -- C++ launchers, kernel handle objects and template instantiations
    # would show up here (round 5 exported ~ 190 of them beside the C entry points)
    return sorted(l.split()[-1] for l in out.splitlines() if l.strip())


def lib_table_versions_agree(_lib):
    """ribca_internal_table(v) is non-NULL exactly for the RIBCA_INTERNAL_VERSION the sources state"""
    text = open(os.path.join(ROOT, "multiplexed-image-annotator_amd", "csrc", "ribca_internal.h")).read()
    v = int(re.search(r"#define RIBCA_INTERNAL_VERSION (\d+)", text).group(1))
    handle = ctypes.CDLL(_lib.LIB_PATH)
    handle.ribca_internal_table.restype = ctypes.c_void_p
    handle.ribca_internal_table.argtypes = [ctypes.c_int32]
    return bool(handle.ribca_internal_table(v)) and not handle.ribca_internal_table(v + 1) and not handle.ribca_internal_table(0)


def test_header_symbols_are_exported_and_bound():
    import __graft_entry__
    __graft_entry__.build()
    from multiplexed_image_annotator_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/ribca_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.SIGNATURES"
    assert sorted(_lib.SIGNATURES) == names
    # the product library exports the product ABI and NOTHING else -- no test hook, no A/B switch, no C++ symbol, no kernel handle
    # (-fvisibility=hidden + csrc/exports.map); the one entry point beyond the reference-facing ones is ribca_internal_table, the versioned
    # launcher table of the hook library, declared in the header as outside the stable ABI
    assert exported_symbols(_lib.LIB_PATH) == names
    assert "ribca_internal_table" in names
    assert _lib.lib().ribca_version() >= 100


def test_test_hooks_live_in_their_own_library():
    """include/ribca_hip_test.h == exports of libribca_hip_test.so == _lib.TEST_SIGNATURES, disjoint from the product ABI; the package never
    names a hook (only tests/ and tools/ load that library)"""
    import __graft_entry__
    __graft_entry__.build()
    from multiplexed_image_annotator_amd import _lib
    hooks = declared_symbols("ribca_hip_test.h")
    assert len(hooks) >= 20 and not set(hooks) & set(declared_symbols())
    assert exported_symbols(_lib.TEST_LIB_PATH) == hooks
    # the hook library needs nothing from the product library but C entry points (it is linked with --no-undefined against them)
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.TEST_LIB_PATH], capture_output=True, text=True, check=True).stdout
    wanted = sorted(l.split()[-1] for l in und.splitlines() if "ribca" in l)
    assert wanted and all(w in declared_symbols() for w in wanted), wanted
    assert lib_table_versions_agree(_lib)
    assert sorted(_lib.TEST_SIGNATURES) == hooks
    lib = _lib.lib()
    assert lib.ribca_gemm_padded_n(100) in (128, 192)      # resolved through the proxy from the second library (no GPU needed)
    with pytest.raises(AttributeError):
        lib.ribca_no_such_entry_point
    pkg = os.path.join(ROOT, "multiplexed-image-annotator_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py") and f != "_lib.py":
            text = open(os.path.join(pkg, f)).read()
            for h in hooks:
                assert h not in text, f"{f} names the test hook {h}"


def test_library_never_ends_the_process():
    """no abort() / exit() / assert in the sources of either library: a refused request is a status (ribca_common.h launch_error)"""
    csrc = os.path.join(ROOT, "multiplexed-image-annotator_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        text = re.sub(r"//.*", "", open(os.path.join(csrc, f)).read())
        for bad in (r"\babort\s*\(", r"\bexit\s*\(", r"\b_exit\s*\(", r"\bassert\s*\(", r"std::terminate"):
            assert not re.search(bad, text), f"{f} can end the calling process: {bad}"


def test_blob_length_matches_reference_parameter_counts():
    from multiplexed_image_annotator_amd import _lib, synth
    # parameter counts of the five classifiers
    expect = {"nerve": 3.03e6, "immune_base": 12.05e6, "struct": 12.05e6, "immune_extended": 21.40e6, "immune_full": 48.07e6}
    for name, (d, c, k) in synth.VIT_CONFIGS.items():
        n = _lib.lib().ribca_vit_blob_len(d, c, k, 12)
        assert abs(n - expect[name]) / expect[name] < 2e-3
        sd = synth.make_vit_state_dict(name, 1, depth=1)
        assert sum(v.numel() for v in sd.values()) == _lib.lib().ribca_vit_blob_len(d, c, k, 1)


def test_product_path_refuses_to_run_without_gpu():
    import torch
    from multiplexed_image_annotator_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.RibcaError):
        _lib.require_gpu()


def test_entry_points_refuse_bad_arguments_with_a_status():
    """the argument checks of the C ABI run before any HIP call: NULL buffers, bad sizes and NULL handles come back as status != 0 with a
    message naming the entry point -- on this GPU-less container as on the box (no dereference, no launch)"""
    from multiplexed_image_annotator_amd import _lib
    lib = _lib.lib()

    def refused(status, name):
        assert status != 0
        msg = lib.ribca_last_error()
        assert msg and name.encode() in msg, (name, msg)

    refused(lib.ribca_mask_minmax(None, 10, None, None), "ribca_mask_minmax")
    refused(lib.ribca_label_table(None, 4, 4, 3, None, None, None), "ribca_label_table")
    refused(lib.ribca_channel_min(None, 3, 16, None, None), "ribca_channel_min")
    refused(lib.ribca_extract_patches(None, 3, 8, 8, None, None, None, None, None, 5, None, None, None), "ribca_extract_patches")
    refused(lib.ribca_vote(None, 3, None, None, 0, None, None, 0.3, 7, None, None, None), "ribca_vote")
    refused(lib.ribca_colorize(None, 16, None, 4, None, None, None, None, None, None, None), "ribca_colorize")
    refused(lib.ribca_knn_cooccurrence(None, None, None, 10, 3, 4, None, None), "ribca_knn_cooccurrence")
    refused(lib.ribca_gauss1d(None, None, 1, 4, 4, 0, None, 1, 0, None), "ribca_gauss1d")
    refused(lib.ribca_radix_hist(None, 1, 16, None, 0, 0, 8, None, None), "ribca_radix_hist")
    refused(lib.ribca_vit_forward(None, None, 3, None, 5, None, None, 0, 4, None), "ribca_vit_forward")
    refused(lib.ribca_vit_forward_precise(None, None, 3, None, 5, None, None, 0, 4, None), "ribca_vit_forward")
    refused(lib.ribca_mae_impute(None, None, None, 3, 5, None, 0, 4, None), "ribca_mae_impute")
    handle = ctypes.c_void_p()
    refused(lib.ribca_vit_create(None, 0, 100, 3, 4, 2, None, ctypes.byref(handle)), "ribca_vit_create")      # D not a multiple of 48
    assert not handle.value
    refused(lib.ribca_mae_create(None, 0, 40, 2, 2, None, ctypes.byref(handle)), "ribca_mae_create")
    assert lib.ribca_vit_workspace_bytes(None, 16) == 0 and lib.ribca_mae_workspace_bytes(None, 16, 3) == 0
    assert lib.ribca_vit_flops_per_cell(None) == 0.0
    lib.ribca_vit_destroy(None)
    lib.ribca_mae_destroy(None)
    assert lib.ribca_vote(None, 3, None, None, 0, None, None, 0.3, 0, None, None, None) == 0      # n = 0: nothing to do, nothing touched
