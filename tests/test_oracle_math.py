"""The oracle's spec'd transcendental functions (Cephes single precision) against float64 libm."""
import ctypes as C

import numpy as np


def _vec(fn, *args):
    fn.restype = C.c_float
    fn.argtypes = [C.c_float] * len(args)
    return np.array([fn(*[float(a[i]) for a in args]) for i in range(len(args[0]))], dtype=np.float32)


def ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.maximum(np.abs(want32), np.float32(1e-30)))
    return np.abs(got.astype(np.float64) - want64) / ulp


def test_sincos(oracle_lib):
    x = np.concatenate([np.linspace(0, 2 * np.pi * 1.0000001, 4001), np.linspace(-100, 100, 2001)]).astype(np.float32)
    s = _vec(oracle_lib.ora_sin, x)
    c = _vec(oracle_lib.ora_cos, x)
    assert np.max(np.abs(s - np.sin(x.astype(np.float64)))) < 2.5e-7
    assert np.max(np.abs(c - np.cos(x.astype(np.float64)))) < 2.5e-7


def test_atan2(oracle_lib):
    rng = np.random.default_rng(1)
    y = rng.uniform(-3, 3, 4000).astype(np.float32)
    x = rng.uniform(-3, 3, 4000).astype(np.float32)
    got = _vec(oracle_lib.ora_atan2, y, x)
    assert np.max(np.abs(got - np.arctan2(y.astype(np.float64), x.astype(np.float64)))) < 6e-7
    z = np.float32(0)
    assert _vec(oracle_lib.ora_atan2, np.array([1, -1, 0, 0], np.float32), np.array([z, z, -1, 1], np.float32)).tolist() == \
        [np.float32(np.pi / 2), np.float32(-np.pi / 2), np.float32(np.pi), 0.0]


def test_exp(oracle_lib):
    x = np.linspace(-87, 20, 5001).astype(np.float32)
    got = _vec(oracle_lib.ora_exp, x)
    want = np.exp(x.astype(np.float64))
    assert np.max(np.abs(got - want) / want) < 3e-7
    assert _vec(oracle_lib.ora_exp, np.array([-200.0, 0.0], np.float32)).tolist() == [0.0, 1.0]
