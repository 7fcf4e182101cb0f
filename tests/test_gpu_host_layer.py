"""GPU test of the C++ drop-in layer: tests/host/host_layer_test renders a scene through PassGenerator / Scene /
Mesh / materials / lights exactly like the viewer would, and the CPU oracle replays the same inputs (the layer's
own baked material rows and light block) — the buffers must be bit-identical."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib
from heatray_amd import _ffi as ffi
from heatray_amd import core, host, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host", "host_layer_test")


def test_host_layer_render_matches_oracle(tmp_path, golden):
    assert os.path.exists(EXE), "tests/host/host_layer_test not built (python -c 'import __graft_entry__ as g; g.build()')"
    W, H, depth, passes = 96, 54, 6, 16   # = maxRenderPasses of the layer test: the last pass delivers the complete image
    sp, sn, suv, si = scenes.uv_sphere(16, 16, 1.0)
    pp, pn, puv, pi = scenes.plane_strip(15, 15)
    meshes = [  # (strip, material, transform, pos, nrm, uv, idx)
        (1, 0, scenes._translate(0, -1.5, 0), pp, pn, puv, pi),
        (0, 1, scenes._translate(-0.9, -0.5, -0.8), sp, sn, suv, si),
        (0, 2, scenes._translate(1.2, -0.5, 0.8), sp, sn, suv, si),
        (0, 3, scenes._translate(0.2, -1.0, 1.8), (sp * np.float32(0.5)).astype(np.float32), sn, suv, si),
    ]
    view = host.orbit_view_matrix(8.0, 0.5, 0.35, target=(0, -0.5, 0))
    scene_file = tmp_path / "scene.bin"
    with open(scene_file, "wb") as f:
        f.write(struct.pack("<4i", W, H, depth, len(meshes)))
        f.write(np.asarray(view, np.float32).T.tobytes())  # column-major
        for strip, mat, xf, p, n, uv, idx in meshes:
            f.write(struct.pack("<2i", strip, mat))
            f.write(np.asarray(xf, np.float32).T.tobytes())
            for arr, dt in ((p, np.float32), (n, np.float32), (uv, np.float32), (idx, np.int32)):
                a = np.ascontiguousarray(arr, dtype=dt).reshape(-1)
                f.write(struct.pack("<i", a.size))
                f.write(a.tobytes())
    out = subprocess.run([EXE, str(scene_file), str(tmp_path), str(passes)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = np.fromfile(tmp_path / "pixels.bin", dtype=np.float32).reshape(H, W, 4)

    # ---- replay through the oracle with the layer's own baked blocks
    gpu = core.create_engine()
    lut, _ = gpu.generate_multiscatter_lut()
    P = 16
    seq = np.stack([gpu.qmc_generate(ffi.HR_SAMPLE_SOBOL, s, P) for s in range(16)])
    ap = np.stack([gpu.qmc_generate(ffi.HR_SAMPLE_SOBOL, s, P, radial=True) for s in range(16)])
    off = gpu.qmc_generate(ffi.HR_SAMPLE_SOBOL, 0, W * H)
    o = oracle_lib.engine()
    o.resize(W, H)
    assert o.create_texture(lut, wrap=ffi.HR_WRAP_CLAMP_TO_EDGE) == 0               # texture 0: multiscatter LUT
    assert o.create_texture(np.full((1, 1, 3), 0.5, np.float32), wrap=ffi.HR_WRAP_CLAMP_TO_EDGE) == 1  # texture 1: solid environment
    rows = np.fromfile(tmp_path / "materials.bin", dtype=np.uint8).reshape(len(meshes), C.sizeof(ffi.Material))
    ids = np.fromfile(tmp_path / "material_ids.bin", dtype=np.int32)
    for row, mid in zip(rows, ids):
        o.set_material(int(mid), ffi.Material.from_buffer_copy(row.tobytes()))
    for (strip, mat, xf, p, n, uv, idx), mid in zip(meshes, ids):
        o.add_mesh(p, n, idx, uvs=uv, mode=ffi.HR_TRIANGLE_STRIP if strip else ffi.HR_TRIANGLES, world=xf, material_id=int(mid))
    o.commit()
    lights = ffi.Lights.from_buffer_copy(open(tmp_path / "lights.bin", "rb").read())
    assert lights.n_directional == 1 and lights.n_point == 1 and lights.n_spot == 1 and lights.env_enabled == 1 and lights.env_texture == 1
    o.set_lights(lights)
    o.set_sequences(seq, ap)
    o.set_seq_offsets(off)
    cam = np.fromfile(tmp_path / "camera.bin", dtype=np.float32)
    opts = host.RenderOptions(max_render_passes=P, max_ray_depth=depth, aspect_ratio=W / H, focus_distance=8.0, fstop=host.FSTOP_DISABLED,
                              view_matrix=view)
    for s in range(passes):
        pp_ = opts.pass_params(s)
        pp_.fov_tan, pp_.aspect_ratio, pp_.focus_distance, pp_.aperture_radius = (float(x) for x in cam)
        o.render_pass(pp_)
    want = o.readback()
    assert (got[..., 3] == passes).all()
    nbad = int((got != want).any(axis=-1).sum())
    assert got.tobytes() == want.tobytes(), f"{nbad} differing pixels"
    # PixelPackBuffer::resolveForDisplay (device-side displayGL.frag) of the same frame, with the test's PostProcessingParams
    shown = np.fromfile(tmp_path / "display.bin", dtype=np.uint8).reshape(H, W, 4)
    P = ffi.DisplayParams.from_buffer_copy(open(tmp_path / "display_params.bin", "rb").read())   # as the layer baked them
    assert P.tonemapping_enabled == 1 and abs(P.camera_exposure - 2.0 ** 0.75) < 1e-6 and abs(P.vignette_falloff - 0.4) < 1e-7
    assert shown.tobytes() == o.display(P, ffi.HR_DISPLAY_RGBA8).tobytes()
