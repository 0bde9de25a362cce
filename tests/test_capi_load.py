"""CPU-side checks of the drop-in boundary: the shared library builds for gfx950, loads, and
exports every entry point include/vgpmp.h declares (no compute call without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

from vgpmp_amd import build, capi

ROOT = Path(__file__).resolve().parent.parent


def test_library_builds_loads_and_exports_header_symbols():
    lib = build.build(force=False, verbose=False)
    assert lib.exists()
    handle = capi.load(require=True)
    header = (ROOT / "include" / "vgpmp.h").read_text()
    declared = set(re.findall(r"\b(vgpmp_[a-z_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(handle, name), name
    assert b"gfx950" in handle.vgpmp_version()


def test_struct_sizes_match_header_layout():
    # sizes implied by include/vgpmp.h (natural alignment, 8-byte tail for the doubles)
    assert ctypes.sizeof(capi.Robot) == 4 * 4 + 7 * 16 * 4 + 12 * 4 + 64 * 4 + 64 * 12 + 64 * 4 + 64 * 4 + 8 + 24 + 64 * 4 + 16 * 32 + 64 * 16 + 64 * 8 + 80
    assert ctypes.sizeof(capi.Sdf) == 8 + 16 + 24 + 8 + 8 + 8 + 16 + 16          # + free_mask, shift / count / words / reserved, clearances
    assert ctypes.sizeof(capi.Dims) == 40
    assert ctypes.sizeof(capi.Params) == 32 and ctypes.sizeof(capi.Noise) == 40
    assert ctypes.sizeof(capi.Problem) == 80 and ctypes.sizeof(capi.Outputs) == 64 and ctypes.sizeof(capi.LikParams) == 72
    assert ctypes.sizeof(capi.InducingParams) == 48


def test_argument_errors_without_gpu():
    handle = capi.load(require=True)
    n = ctypes.c_size_t(0)
    bad = capi.Dims(1, 8, 8, 10, 47, 7, 64, 1, 0, 0)          # M + 2 > VGPMP_MAX_MZ
    assert handle.vgpmp_workspace_bytes(ctypes.byref(bad), ctypes.byref(n)) == -2
    bad = capi.Dims(1, 8, 8, 10, 5, 7, 60, 1, 0, 0)           # B not a multiple of 16
    assert handle.vgpmp_workspace_bytes(ctypes.byref(bad), ctypes.byref(n)) == -2
    ok = capi.Dims(1, 128, 128, 100, 30, 7, 1024, 4, 0, 0)
    assert handle.vgpmp_workspace_bytes(ctypes.byref(ok), ctypes.byref(n)) == 0 and n.value > 0
    assert handle.vgpmp_workspace_bytes(None, ctypes.byref(n)) == -1
    # voxel-table sizes: linear = 16 B per voxel, bricked = extents rounded up to whole 4x4x4 bricks + 4 B per brick
    tb, bb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert handle.vgpmp_sdf_table_bytes(5, 1, 9, capi.SDF_LINEAR, ctypes.byref(tb), ctypes.byref(bb)) == 0
    assert (tb.value, bb.value) == (5 * 9 * 16, 0)
    assert handle.vgpmp_sdf_table_bytes(5, 1, 9, capi.SDF_BRICK4, ctypes.byref(tb), ctypes.byref(bb)) == 0
    assert (tb.value, bb.value) == (2 * 1 * 3 * 1024, 2 * 1 * 3 * 4)
    assert handle.vgpmp_sdf_table_bytes(512, 512, 512, capi.SDF_BRICK4, ctypes.byref(tb), ctypes.byref(bb)) == 0
    assert (tb.value, bb.value) == (2 << 30, 8 << 20)
    assert handle.vgpmp_sdf_table_bytes(4, 4, 4, 7, ctypes.byref(tb), ctypes.byref(bb)) == -1
    assert handle.vgpmp_sdf_table_bytes(0, 4, 4, 0, ctypes.byref(tb), ctypes.byref(bb)) == -2
    # free-space masks: one bit per block of 2^shift voxels per edge, words rounded up to 16 bytes
    w = ctypes.c_size_t(0)
    assert handle.vgpmp_sdf_mask_words(512, 512, 512, 3, ctypes.byref(w)) == 0 and w.value * 4 == 32 << 10
    assert handle.vgpmp_sdf_mask_words(130, 154, 80, 2, ctypes.byref(w)) == 0 and w.value == ((33 * 39 * 20 + 31) // 32 + 3) // 4 * 4
    assert handle.vgpmp_sdf_mask_words(8, 8, 8, 1, ctypes.byref(w)) == -2
    assert handle.vgpmp_sdf_free_mask(None, None) == -1


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    monkeypatch.setenv("VGPMP_HIP_LIB", str(tmp_path / "missing.so"))
    monkeypatch.setattr(capi, "_lib", None)
    with pytest.raises(capi.VgpmpError):
        capi.load(require=True)
