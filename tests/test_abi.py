"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol that
include/monogs_raster.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "monogs_raster.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mgs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = _declared_symbols()
    for s in ("mgs_forward_preprocess", "mgs_forward_render", "mgs_backward", "mgs_mark_visible", "mgs_dist2_knn",
              "mgs_geometry_bytes", "mgs_image_bytes", "mgs_binning_bytes", "mgs_backward_bytes",
              "mgs_knn_scratch_bytes", "mgs_abi_version", "mgs_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol(native_lib):
    from monogs_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(raw, s), f"{s} declared in the header but not exported"
    assert set(_lib.SIGNATURES) == set(_declared_symbols())
    assert native_lib.mgs_abi_version() == _lib.ABI_VERSION


def test_scratch_size_functions_are_pure(native_lib):
    assert native_lib.mgs_geometry_bytes(0) >= 256
    a, b = native_lib.mgs_geometry_bytes(1000), native_lib.mgs_geometry_bytes(2000)
    assert b > a >= 1000 * 64
    assert native_lib.mgs_image_bytes(640, 480) >= 640 * 480 * 8 + 1200 * 8
    assert native_lib.mgs_backward_bytes(1000) >= 1000 * 64
    assert native_lib.mgs_knn_scratch_bytes(10) > 0


def test_struct_layout_matches_header():
    from monogs_amd import _lib
    assert ctypes.sizeof(_lib.MgsCamera) == 9 * 4 + 4 + 5 * 8       # nine 4-byte fields, padding, five pointers
    assert ctypes.sizeof(_lib.MgsTiming) == 9 * 4


def test_code_object_targets_gfx950(native_lib):
    from monogs_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_product_path_has_no_cpu_fallback():
    """Nothing under monogs_amd/ may import the oracle, and CPU tensors are rejected."""
    import torch

    for dirpath, _, files in os.walk(os.path.join(ROOT, "monogs_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
    from monogs_amd.knn import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(torch.zeros(10, 3))


def test_maps_beyond_the_gradient_line_offset_are_refused(native_lib):
    """The blend backward addresses a gradient line as index * 64 + slot in 32 bits (csrc/blend.hip, bt_flush): P >= 2^26 must
    come back as an argument error from every entry point that takes a map, before anything is launched (no GPU needed)."""
    from monogs_amd import _lib
    text = open(os.path.join(ROOT, "include", "monogs_raster.h")).read()
    assert re.search(r"#define MGS_MAX_GAUSSIANS \(\(1 << 26\) - 1\)", text)
    cam = _lib.MgsCamera()
    cam.image_height, cam.image_width = 16, 16
    for f in ("bg", "viewmatrix", "projmatrix", "projmatrix_raw", "campos"):
        setattr(cam, f, 0x1000)                      # non-NULL and never dereferenced: the size check comes first
    for name in ("mgs_forward_preprocess", "mgs_forward_capacity", "mgs_forward_render", "mgs_forward_render_capacity",
                 "mgs_backward"):
        args = []
        for k, t in enumerate(_lib.SIGNATURES[name][1]):
            if k == 0:
                args.append(ctypes.byref(cam))
            elif k == 1:
                args.append(1 << 26)                 # P
            elif t in (ctypes.c_uint64, ctypes.c_int32):
                args.append(1)
            else:
                args.append(None)
        assert getattr(native_lib, name)(*args) == 1, name
        assert b"MGS_MAX_GAUSSIANS" in native_lib.mgs_last_error(), (name, native_lib.mgs_last_error())


def test_header_marks_the_process_global_debug_setters():
    """`include/monogs_raster.h` promises a library without state between calls; the debug setters are the declared exception
    (VERDICT round 4, item 8): each one that writes a process-global must say so where it is declared -- test use only, process-global,
    not thread-safe -- so that nobody reads the promise as covering them."""
    import re
    text = open(os.path.join(ROOT, "include", "monogs_raster.h")).read()
    for name in ("mgs_debug_set_radix_spin_limit", "mgs_debug_set_option", "mgs_debug_set_blend_events"):
        m = re.search(r"/\*((?:(?!\*/).)*)\*/\s*int %s\(" % name, text, re.S)
        assert m, name
        comment = " ".join(m.group(1).split()).lower()
        assert "use only" in comment and "process-global" in comment and "not thread-safe" in comment, (name, comment[:200])
    # and no OTHER setter of global state hides in the header
    setters = set(re.findall(r"int (mgs_debug_set_\w+)\(", text))
    assert setters == {"mgs_debug_set_radix_spin_limit", "mgs_debug_set_option", "mgs_debug_set_blend_events"}, setters
