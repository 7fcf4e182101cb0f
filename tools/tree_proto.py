#!/usr/bin/env python3
"""tools/tree_proto.py {soup|terrain} N — CPU prototype (numpy) behind round 4's builder decision (DESIGN.md §2 "Tree quality").

What could a better binary tree buy?  For rays distributed like uniform random lines the expected number of 4-wide nodes a ray visits
is (sum of the node boxes' surface areas) / (root area) — the quantity the product's collapse already minimises for a GIVEN binary
tree (hr_build.hip: k_refit_round's DP costs).  This script builds, for the benchmark's triangle soup and for a terrain mesh,
  * the LBVH the product builds (30-bit Morton codes, Karras splits),
  * PLOC (Meister & Bittner 2018) with search radius r,
  * the LBVH with its bottom subtrees (<= T triangles) rebuilt agglomeratively or by full-sweep SAH,
  * a full-sweep SAH build of everything (an upper bound no GPU builder reaches),
collapses each with the same DP and prints that sum.  Output of round 4: profiles/r4_tree_proto.txt.
"""
import numpy as np, sys, time
rng = np.random.default_rng(1)

def soup(N):
    l = 0.5 * N ** (-1/3)
    c = rng.uniform(-1, 1, (N, 3))
    e1 = rng.uniform(-l, l, (N, 3)); e2 = rng.uniform(-l, l, (N, 3))
    v = np.stack([c, c + e1, c + e2], 1)
    return v.min(1), v.max(1)

def terrain(nx, ny):
    # grid mesh, heights from smooth noise
    x = np.linspace(-1, 1, nx + 1); y = np.linspace(-0.5, 0.5, ny + 1)
    X, Y = np.meshgrid(x, y, indexing='ij')
    Z = 0.15*np.sin(5*X)*np.cos(7*Y) + 0.05*np.sin(23*X+1.3)*np.sin(19*Y) + 0.01*rng.normal(size=X.shape)
    P = np.stack([X, Z, Y], -1)
    a = P[:-1, :-1]; b = P[1:, :-1]; c = P[:-1, 1:]; d = P[1:, 1:]
    t1 = np.stack([a, b, c], -2).reshape(-1, 3, 3); t2 = np.stack([b, d, c], -2).reshape(-1, 3, 3)
    t = np.concatenate([t1, t2])
    return t.min(1), t.max(1)

def morton(lo, hi):
    c = (lo + hi) * 0.5
    mn, mx = lo.min(0), hi.max(0)
    q = np.clip(((c - mn) / (mx - mn) * 1024).astype(np.int64), 0, 1023)
    def exp(v):
        v = (v * 0x00010001) & 0xFF0000FF
        v = (v * 0x00000101) & 0x0F00F00F
        v = (v * 0x00000011) & 0xC30C30C3
        v = (v * 0x00000005) & 0x49249249
        return v
    return (exp(q[:, 0]) << 2) | (exp(q[:, 1]) << 1) | exp(q[:, 2])

def area(lo, hi):
    d = hi - lo
    return d[..., 0]*d[..., 1] + d[..., 1]*d[..., 2] + d[..., 2]*d[..., 0]

def lbvh(keys):
    """children arrays in an order where children precede parents; leaf i = ~i"""
    n = len(keys)
    # combine key with index to make unique
    k = (keys.astype(np.int64) << 32) | np.arange(n)
    left = []; right = []
    # iterative top-down; produce nodes then reverse
    nodes = []  # (lo, hi, parent, side)
    stack = [(0, n - 1, -1, 0)]
    L = []; R = []
    while stack:
        lo, hi, par, side = stack.pop()
        if lo == hi:
            ref = ~lo
        else:
            ref = len(L); L.append(0); R.append(0)
            x = k[lo] ^ k[hi]
            bit = x.bit_length() - 1 if isinstance(x, int) else int(x).bit_length() - 1
            # split: first index with bit set
            mask = 1 << bit
            a, b = lo, hi
            # binary search first element with bit set (sorted => prefix equal above)
            while a < b:
                m = (a + b) // 2
                if int(k[m]) & mask: b = m
                else: a = m + 1
            split = a  # first with bit set
            stack.append((lo, split - 1, ref, 0)); stack.append((split, hi, ref, 1))
        if par >= 0:
            if side == 0: L[par] = ref
            else: R[par] = ref
    L = np.array(L); R = np.array(R)
    # order: parents before children in index; process in reverse
    return L, R, list(range(len(L) - 1, -1, -1)), 0

def ploc(lo, hi, r):
    n = len(lo)
    clo, chi = lo.copy(), hi.copy()
    ref = ~np.arange(n)          # current cluster refs
    L = []; R = []; nlo = []; nhi = []
    nnodes = 0
    it = 0
    while len(ref) > 1:
        m = len(ref)
        best = np.full(m, np.inf); nn = np.full(m, -1)
        idx = np.arange(m)
        for off in list(range(-r, 0)) + list(range(1, r + 1)):
            j = idx + off
            ok = (j >= 0) & (j < m)
            jj = np.clip(j, 0, m - 1)
            a = area(np.minimum(clo, clo[jj]), np.maximum(chi, chi[jj]))
            a = np.where(ok, a, np.inf)
            upd = a < best
            best = np.where(upd, a, best); nn = np.where(upd, jj, nn)
        mutual = (nn[nn] == idx) & (idx < nn)
        i = idx[mutual]; j = nn[mutual]
        k = len(i)
        newref = nnodes + np.arange(k)
        L.append(ref[i]); R.append(ref[j])
        mlo = np.minimum(clo[i], clo[j]); mhi = np.maximum(chi[i], chi[j])
        nlo.append(mlo); nhi.append(mhi)
        nnodes += k
        ref = ref.copy(); ref[i] = newref; clo[i] = mlo; chi[i] = mhi
        keep = np.ones(m, bool); keep[j] = False
        ref, clo, chi = ref[keep], clo[keep], chi[keep]
        it += 1
    L = np.concatenate(L); R = np.concatenate(R)
    return L, R, list(range(len(L))), len(L) - 1, it

def evaluate(L, R, order, root, lo, hi, label):
    """order: node ids with children before parents.  DP cost of the collapse to 4-wide (areas), plus binary SAH sum and depth"""
    m = len(L)
    blo = np.zeros((m, 3)); bhi = np.zeros((m, 3))
    cost = np.zeros((m, 4)); depth = np.zeros(m, int)
    for i in order:
        l, r = L[i], R[i]
        alo, ahi = (lo[~l], hi[~l]) if l < 0 else (blo[l], bhi[l])
        clo_, chi_ = (lo[~r], hi[~r]) if r < 0 else (blo[r], bhi[r])
        blo[i] = np.minimum(alo, clo_); bhi[i] = np.maximum(ahi, chi_)
        cl = np.zeros(4) if l < 0 else cost[l]; cr = np.zeros(4) if r < 0 else cost[r]
        h2 = cl[0] + cr[0]; h3 = min(cl[0] + cr[1], cl[1] + cr[0]); h4 = min(cl[0] + cr[2], cl[1] + cr[1], cl[2] + cr[0])
        c1 = area(blo[i], bhi[i]) + h4
        c2 = min(c1, h2); c3 = min(c2, h3); c4 = min(c3, h4)
        cost[i] = (c1, c2, c3, c4)
        depth[i] = 1 + max(0 if l < 0 else depth[l], 0 if r < 0 else depth[r])
    ra = area(blo[root], bhi[root])
    binsum = area(blo, bhi).sum() / ra
    print(f"{label:28s} 4-wide node area sum / root area = {cost[root,0]/ra:8.3f}   binary sum = {binsum:8.3f}   binary depth = {depth[root]}")
    return cost[root, 0] / ra

def agglo(ids, lo, hi, L, R, blo_list):
    """full agglomerative clustering of leaf ids; appends nodes to L,R; returns ref of the subtree root"""
    refs = [~i for i in ids]
    blo = [lo[i] for i in ids]; bhi = [hi[i] for i in ids]
    while len(refs) > 1:
        m = len(refs)
        B0 = np.array(blo); B1 = np.array(bhi)
        u0 = np.minimum(B0[:, None], B0[None]); u1 = np.maximum(B1[:, None], B1[None])
        a = area(u0, u1); a[np.arange(m), np.arange(m)] = np.inf
        i, j = np.unravel_index(np.argmin(a), a.shape)
        if i > j: i, j = j, i
        L.append(refs[i]); R.append(refs[j])
        refs[i] = len(L) - 1; blo[i] = u0[i, j]; bhi[i] = u1[i, j]
        del refs[j]; del blo[j]; del bhi[j]
    return refs[0]

def sah_sweep(ids, lo, hi, L, R):
    """top-down full-sweep SAH build (binary, to single prims) of leaf ids; returns ref"""
    if len(ids) == 1: return ~ids[0]
    ids = np.array(ids)
    best = (np.inf, None, None)
    c = (lo[ids] + hi[ids]) * 0.5
    for ax in range(3):
        o = np.argsort(c[:, ax]); s = ids[o]
        l0 = np.minimum.accumulate(lo[s], 0); l1 = np.maximum.accumulate(hi[s], 0)
        r0 = np.minimum.accumulate(lo[s][::-1], 0)[::-1]; r1 = np.maximum.accumulate(hi[s][::-1], 0)[::-1]
        n = len(s)
        k = np.arange(1, n)
        cost = area(l0[k-1], l1[k-1]) * k + area(r0[k], r1[k]) * (n - k)
        j = np.argmin(cost)
        if cost[j] < best[0]: best = (cost[j], s, j + 1)
    _, s, k = best
    a = sah_sweep(list(s[:k]), lo, hi, L, R); b = sah_sweep(list(s[k:]), lo, hi, L, R)
    L.append(a); R.append(b)
    return len(L) - 1

def hybrid(keys, lo, hi, T, mode):
    L0, R0, order0, root0 = lbvh(keys)
    m = len(L0)
    cnt = np.zeros(m, int); first = np.zeros(m, int)
    for i in order0:
        l, r = L0[i], R0[i]
        cl = 1 if l < 0 else cnt[l]; cr = 1 if r < 0 else cnt[r]
        cnt[i] = cl + cr
        first[i] = ~l if l < 0 else first[l]
    L = []; R = []
    def build(ref):
        if ref < 0: return ref
        if cnt[ref] <= T:
            ids = list(range(first[ref], first[ref] + cnt[ref]))
            return agglo(ids, lo, hi, L, R, None) if mode == "agglo" else sah_sweep(ids, lo, hi, L, R)
        a = build(L0[ref]); b = build(R0[ref])
        L.append(a); R.append(b)
        return len(L) - 1
    sys.setrecursionlimit(100000)
    root = build(root0)
    return np.array(L), np.array(R), list(range(len(L))), root


if __name__ == "__main__":
    which = sys.argv[1]; N = int(sys.argv[2])
    lo, hi = soup(N) if which == "soup" else terrain(int((N/2)**0.5*1.414), int((N/2)**0.5/1.414))
    pad = 1e-5 * np.linalg.norm(hi.max(0) - lo.min(0))
    lo, hi = lo - pad, hi + pad
    keys = morton(lo, hi)
    o = np.argsort(keys, kind='stable'); lo, hi, keys = lo[o], hi[o], keys[o]
    print(which, len(lo), "prims")
    sys.setrecursionlimit(100000)
    L, R, order, root = lbvh(keys)
    base = evaluate(L, R, order, root, lo, hi, "LBVH")
    for r in (8, 16, 32):
        L, R, order, root, it = ploc(lo, hi, r)
        v = evaluate(L, R, order, root, lo, hi, f"PLOC r={r} ({it} iterations)")
        print(f"   -> {100*(v/base-1):+.1f} %")
    for mode in ("agglo", "sah"):
        for T in (8, 16, 64):
            L, R, order, root = hybrid(keys, lo, hi, T, mode)
            v = evaluate(L, R, order, root, lo, hi, f"LBVH top + {mode} subtrees <= {T}")
            print(f"   -> {100*(v/base-1):+.1f} %")
    L = []; R = []
    root = sah_sweep(list(range(len(lo))), lo, hi, L, R)
    v = evaluate(np.array(L), np.array(R), list(range(len(L))), root, lo, hi, "full-sweep SAH (upper bound)")
    print(f"   -> {100*(v/base-1):+.1f} %")
