"""tests/golden/make_phantoms.py — writes tests/golden/phantom_hits.npz: sliver triangles and rays for which float32 Möller–Trumbore,
in the contract's operation order, ACCEPTS a ray that passes the triangle centimetres away (a 'phantom hit': the determinant of three
nearly collinear vertices is a difference of nearly equal products).  Found by a seeded random search in numpy float32; every operation
below is rounded to float32 exactly as hr_trace.h / oracle_bvh.cpp evaluate it.  The fixture pins the hit test's second half
(hitInTriBox): with it, these rays hit nothing — for brute force, for every tree and on the GPU alike."""
import os
import numpy as np

f = np.float32


def cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1)


def dot(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]


def moller_trumbore(p0, p1, p2, o, d):
    """(accepted, t, u, v) per row, float32, the contract's operation order (hr_trace.h::traverse)."""
    e1, e2 = p1 - p0, p2 - p0
    with np.errstate(all="ignore"):
        pvec = cross(d, e2)
        det = dot(e1, pvec)
        inv = f(1.0) / det
        tvec = o - p0
        u = dot(tvec, pvec) * inv
        qvec = cross(tvec, e1)
        v = dot(d, qvec) * inv
        t = dot(e2, qvec) * inv
    ok = (det != 0) & (u >= 0) & ~(u > 1) & (v >= 0) & ~(u + v > 1)
    return ok, t, u, v


def main(want=24, seed=1):
    rng = np.random.default_rng(seed)
    rows = []
    n = 4_000_000
    while len(rows) < want:
        p0 = rng.uniform(-1, 1, (n, 3)).astype(f)
        e1 = (rng.normal(size=(n, 3)) * 0.05).astype(f)
        s = rng.uniform(0.2, 0.9, (n, 1)).astype(f)
        p1 = (p0 + e1).astype(f)
        p2 = (p0 + e1 * s + rng.normal(size=(n, 3)).astype(f) * f(3e-8)).astype(f)   # three (nearly) collinear vertices
        o = rng.normal(size=(n, 3)).astype(f)
        o = (o / np.linalg.norm(o, axis=1, keepdims=True) * f(3.0)).astype(f)          # a camera's distance
        tgt = (p0 + e1 * f(0.4) + rng.normal(size=(n, 3)).astype(f) * f(0.1)).astype(f)
        d = tgt - o
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f)
        ok, t, u, v = moller_trumbore(p0, p1, p2, o, d)
        ok &= (t > 1e-3) & (t < 1e30)
        P = o + t[:, None] * d
        lo, hi = np.minimum(np.minimum(p0, p1), p2), np.maximum(np.maximum(p0, p1), p2)
        dist = np.maximum(np.maximum(lo - P, P - hi), 0).max(axis=1)
        for i in np.nonzero(ok & (dist > 0.01))[0]:
            rows.append(np.concatenate([p0[i], p1[i], p2[i], o[i], d[i], [t[i], u[i], v[i], dist[i]]]))
    a = np.array(rows[:want], dtype=f)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "phantom_hits.npz")
    np.savez(out, p0=a[:, 0:3], p1=a[:, 3:6], p2=a[:, 6:9], origin=a[:, 9:12], direction=a[:, 12:15], t=a[:, 15], u=a[:, 16], v=a[:, 17], distance=a[:, 18])
    print(out, a.shape)


if __name__ == "__main__":
    main()
