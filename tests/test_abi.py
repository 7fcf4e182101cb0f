"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly what
include/hrcore.h declares (no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

from heatray_amd import _ffi as ffi
from heatray_amd import core

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "hrcore.h")).read()
    return sorted(set(re.findall(r"\b(hr_[a-z0-9_]+)\s*\(", text)))


def test_header_and_python_binding_agree():
    assert sorted("hr_" + s for s in ffi.ABI_SYMBOLS) == declared_functions()


def test_library_exports_every_declared_symbol():
    lib = core.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_oracle_mirrors_the_abi(oracle_lib):
    for s in ffi.ABI_SYMBOLS:
        if s in ("ctx_set_stream", "frame_bind_external", "frame_device_ptr", "get_kernel_times"):
            continue  # device-memory plumbing has no CPU counterpart
        assert hasattr(oracle_lib, "ora_" + s), s


def test_struct_sizes_match_the_c_layout():
    # sizes the C compiler produces for include/hrcore.h (x86-64 SysV); guards the ctypes mirrors
    assert ctypes.sizeof(ffi.CtxDesc) == 32
    assert ctypes.sizeof(ffi.MeshDesc) == 6 * 8 + 7 * 4 + 4 + 8 + 4 + 4 + 64 + 12 + 4
    assert ctypes.sizeof(ffi.Material) == 4 * 10 + 4 * 15
    assert ctypes.sizeof(ffi.Lights) == 4 + 120 + 4 + 120 + 4 + 180 + 40 + 16
    assert ctypes.sizeof(ffi.PassParams) == 7 * 4 + 64 + 4 + 16 + 4 + 7 * 4
    assert ctypes.sizeof(ffi.PassStats) == 80
    assert ctypes.sizeof(ffi.Hit) == 16


def test_no_device_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ffi.EngineError):
        core.create_engine()
