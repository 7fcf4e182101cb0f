"""Known-answer tests of the oracle's display resolve (displayGL.frag:74-151 restated in oracle/oracle_display.cpp).
The reference holds no fixture for its display shader ("parity unpinned"); these pin the closed forms the shader is made of."""
import numpy as np
import pytest

import oracle_lib
from heatray_amd import _ffi as ffi


def frame_engine(rgba):
    """An oracle context whose accumulation buffer holds `rgba` [H, W, 4]: rendered as an empty scene + env colour is
    not flexible enough, so the buffer is filled through the white-furnace trick: one pass over an empty scene with a
    constant environment gives (c, c, c, 1) everywhere; arbitrary buffers come from ora_readback's pointer instead."""
    import ctypes as C
    h, w = rgba.shape[:2]
    eng = oracle_lib.engine()
    eng.resize(w, h)
    p = ffi.f32p()
    ww, hh = C.c_int32(), C.c_int32()
    eng._call("readback", C.byref(p), C.byref(ww), C.byref(hh))   # the oracle hands out its own buffer: write into it
    np.ctypeslib.as_array(p, shape=(h, w, 4))[...] = rgba.astype(np.float32)
    return eng


def srgb(c):
    c = np.asarray(c, dtype=np.float64)
    return np.where(c <= 0.0031308, 12.92 * c, 1.055 * np.power(np.maximum(c, 1e-300), 1 / 2.4) - 0.055)


def srgb_inv(c):
    c = np.asarray(c, dtype=np.float64)
    return np.where(c <= 0.04045, c / 12.92, np.power((c + 0.055) / 1.055, 2.4))


def test_neutral_settings_are_divide_and_srgb_encode():
    rng = np.random.default_rng(7)
    a = rng.integers(1, 64, size=(24, 40, 1)).astype(np.float32)
    rgb = rng.uniform(0.0, 1.0, size=(24, 40, 3)).astype(np.float32)
    eng = frame_engine(np.concatenate([rgb * a, a], axis=-1))
    out = eng.display(ffi.display_params(), ffi.HR_DISPLAY_RGBA32F)
    want = srgb((rgb * a).astype(np.float32) / a)
    # RGB -> HSV -> RGB with neutral factors is the identity up to float rounding
    assert np.abs(out[..., :3] - want).max() < 2e-6 + 3e-6
    assert (out[..., 3] == 1.0).all()
    b8 = eng.display(ffi.display_params(), ffi.HR_DISPLAY_RGBA8)
    assert b8.dtype == np.uint8 and (b8[..., 3] == 255).all()
    assert np.abs(b8[..., :3].astype(np.int32) - np.floor(np.clip(want, 0, 1) * 255 + 0.5).astype(np.int32)).max() <= 1
    assert (b8[..., :3] == np.floor(np.clip(out[..., :3], 0, 1) * 255 + np.float32(0.5)).astype(np.uint8)).all()


def test_pow_log_restatement_accuracy():
    # the pow of the pipeline (exp(y log x)) against float64 over the display range
    x = np.concatenate([np.linspace(1e-4, 1.0, 4000), np.linspace(1.0, 64.0, 1000)]).astype(np.float32)
    eng = frame_engine(np.stack([x, x, x, np.ones_like(x)], axis=-1).reshape(1, -1, 4))
    out = eng.display(ffi.display_params(), ffi.HR_DISPLAY_RGBA32F)[0, :, 0]
    ref = srgb(x.astype(np.float64))
    assert np.abs(out - ref).max() / 1.0 < 4e-6 * ref.max()


def test_hdr_format_and_unsampled_pixels():
    buf = np.zeros((4, 4, 4), np.float32)
    buf[1, 2] = (2.0, 4.0, 8.0, 4.0)
    eng = frame_engine(buf)
    hdr = eng.display(ffi.display_params(), ffi.HR_DISPLAY_HDR_RGBA32F)
    assert tuple(hdr[1, 2]) == (0.5, 1.0, 2.0, 4.0)               # saveScreenshot: rgb * (1 / a), a kept
    assert (hdr[0, 0] == 0).all()                                   # no samples: black, not NaN
    b8 = eng.display(ffi.display_params(), ffi.HR_DISPLAY_RGBA8)
    assert tuple(b8[0, 0]) == (0, 0, 0, 255)
    assert tuple(b8[1, 2]) == (188, 255, 255, 255)                  # srgb(0.5) = 0.7354 -> 188


def test_aces_tonemap_exposure_and_levels():
    v = np.linspace(0.0, 8.0, 64, dtype=np.float32)
    eng = frame_engine(np.stack([v, v, v, np.ones_like(v)], axis=-1).reshape(1, -1, 4))
    tm = eng.display(ffi.display_params(tonemapping_enabled=True), ffi.HR_DISPLAY_RGBA32F)[0, :, 0]
    # closed form of the shader for grey input: ACES matrices rows sum to ~1 (grey stays grey within 1e-3)
    s = srgb(v.astype(np.float64))
    x = s * (0.59719 + 0.35458 + 0.04823)
    fit = (x * (x + 0.0245786) - 0.000090537) / (x * (0.983729 * x + 0.4329510) + 0.238081)
    y = np.clip(fit * (1.60475 - 0.53108 - 0.07367), 0, 1)
    want = srgb(srgb_inv(y))
    assert np.abs(tm - want).max() < 2e-5
    assert (np.diff(tm) >= -1e-6).all() and tm.max() <= 1.0 + 1e-6           # monotone, bounded
    # exposure +1 stop doubles linear radiance; red/green/blue scale channels
    out = eng.display(ffi.display_params(exposure=1.0, red=0.5, blue=0.0), ffi.HR_DISPLAY_RGBA32F)[0]
    assert np.abs(out[:, 1] - srgb(2.0 * v.astype(np.float64))).max() < 3e-5 * 8
    assert np.abs(out[:, 0] - srgb(1.0 * v.astype(np.float64))).max() < 3e-5 * 8
    assert (out[:, 2] == 0).all()


def test_vignette_and_saturation():
    ones = np.ones((33, 33, 4), np.float32) * np.array([0.2, 0.5, 0.8, 1.0], np.float32)
    eng = frame_engine(ones)
    out = eng.display(ffi.display_params(vignette_intensity=1.0, vignette_falloff=0.2), ffi.HR_DISPLAY_RGBA32F)
    centre, corner = out[16, 16, :3], out[0, 0, :3]
    assert np.abs(centre - srgb([0.2, 0.5, 0.8])).max() < 1e-5                # smoothstep(0.8, 0.16, 0) = 1 at the centre
    assert (corner < centre).all()                                              # darker towards the corners
    # reversed-edge smoothstep as the shader writes it: distance * (intensity + blue)
    d = np.hypot(0.5 - 0.5 / 33, 0.5 - 0.5 / 33) * 2.0
    t = np.clip((d - 0.8) / (0.2 * 0.799 - 0.8), 0, 1)
    assert np.abs(corner - srgb(np.array([0.2, 0.5, 0.8]) * (t * t * (3 - 2 * t)))).max() < 1e-5
    grey = eng.display(ffi.display_params(saturation=0.0), ffi.HR_DISPLAY_RGBA32F)[5, 5, :3]
    assert np.abs(grey - srgb(0.8)).max() < 1e-5                                 # saturation 0: value (max channel) only


def test_display_of_a_shard_writes_only_owned_pixels():
    buf = np.ones((70, 100, 4), np.float32)
    from heatray_amd import tiles
    import ctypes as C
    eng = oracle_lib.engine(rank=1, world=3, tile_size=32)
    eng.resize(100, 70)
    p = ffi.f32p()
    eng._call("readback", C.byref(p), None, None)
    np.ctypeslib.as_array(p, shape=(70, 100, 4))[...] = buf
    b8 = eng.display(ffi.display_params(), ffi.HR_DISPLAY_RGBA8)
    own = tiles.owner_map(100, 70, 3) == 1
    assert (b8[own] == 255).all() and (b8[~own] == 0).all()
