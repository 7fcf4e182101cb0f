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


def test_abi_version_of_header_binding_library_and_oracle_agree(oracle_lib):
    text = open(os.path.join(ROOT, "include", "hrcore.h")).read()
    (ver,) = re.findall(r"#define HR_ABI_VERSION (\d+)u", text)
    assert int(ver) == ffi.HR_ABI_VERSION
    for lib, name in ((core.load_library(), "hr_abi_version"), (oracle_lib, "ora_abi_version")):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_uint32
        assert fn() == ffi.HR_ABI_VERSION, name


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
    assert ctypes.sizeof(ffi.CtxDesc) == 40
    assert ctypes.sizeof(ffi.MeshDesc) == 6 * 8 + 7 * 4 + 4 + 8 + 4 + 4 + 64 + 12 + 4
    assert ctypes.sizeof(ffi.Material) == 4 * 10 + 4 * 15
    assert ctypes.sizeof(ffi.Lights) == 4 + 120 + 4 + 120 + 4 + 180 + 40 + 16
    assert ctypes.sizeof(ffi.PassParams) == 7 * 4 + 64 + 4 + 16 + 4 + 7 * 4
    assert ctypes.sizeof(ffi.PassStats) == 80
    assert ctypes.sizeof(ffi.Hit) == 16


def test_every_mirrored_struct_has_the_size_and_field_offsets_gcc_gives_the_header(tmp_path):
    # the ctypes mirrors against the header itself: a C program prints sizeof and the offset of every field the mirror names
    import subprocess
    pairs = [("hr_ctx_desc", ffi.CtxDesc), ("hr_mesh_desc", ffi.MeshDesc), ("hr_scene_info", ffi.SceneInfo), ("hr_texture_desc", ffi.TextureDesc),
             ("hr_material", ffi.Material), ("hr_lights", ffi.Lights), ("hr_pass_params", ffi.PassParams), ("hr_pass_stats", ffi.PassStats),
             ("hr_kernel_times", ffi.KernelTimes), ("hr_step_record", ffi.StepRecord), ("hr_display_params", ffi.DisplayParams), ("hr_hit", ffi.Hit)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hrcore.h"', "int main(void) {"]
    for cname, mirror in pairs:
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in mirror._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True, text=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, mirror in pairs:
        assert int(got[cname]) == ctypes.sizeof(mirror), cname
        for fname, _ in mirror._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(mirror, fname).offset, (cname, fname)


def test_no_device_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ffi.EngineError):
        core.create_engine()
