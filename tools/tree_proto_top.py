"""tools/tree_proto_top.py {soup|terrain} N — companion of tree_proto.py: keep the bottom subtrees (<= T triangles) of the PLOC or the radix tree and
rebuild everything above them with a full-sweep SAH over the subtree boxes (what a top-level SAH builder could add).  Output: profiles/r4_tree_proto.txt."""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tools')
from tree_proto import *

def leafcount(L, R, order):
    cnt = np.zeros(len(L), int)
    for i in order:
        cnt[i] = (1 if L[i] < 0 else cnt[L[i]]) + (1 if R[i] < 0 else cnt[R[i]])
    return cnt

def sah_top(L, R, order, root, lo, hi, T):
    """keep subtrees with <= T leaves of the given tree, rebuild everything above them by full-sweep SAH over the subtree boxes"""
    m = len(L)
    cnt = leafcount(L, R, order)
    blo = np.zeros((m, 3)); bhi = np.zeros((m, 3))
    for i in order:
        l, r = L[i], R[i]
        a0, a1 = (lo[~l], hi[~l]) if l < 0 else (blo[l], bhi[l])
        b0, b1 = (lo[~r], hi[~r]) if r < 0 else (blo[r], bhi[r])
        blo[i] = np.minimum(a0, b0); bhi[i] = np.maximum(a1, b1)
    # cut
    cut = []
    stack = [root]
    while stack:
        x = stack.pop()
        if x < 0 or cnt[x] <= T: cut.append(x); continue
        stack.append(L[x]); stack.append(R[x])
    clo = np.array([lo[~c] if c < 0 else blo[c] for c in cut]); chi = np.array([hi[~c] if c < 0 else bhi[c] for c in cut])
    # weights: SAH with leaf counts as primitive counts
    w = np.array([1 if c < 0 else cnt[c] for c in cut])
    Ln = list(L); Rn = list(R)
    def build(ids):
        if len(ids) == 1: return cut[ids[0]]
        ids = np.array(ids); best = (np.inf, None, None)
        c = (clo[ids] + chi[ids]) * 0.5
        for ax in range(3):
            o = np.argsort(c[:, ax]); s = ids[o]
            l0 = np.minimum.accumulate(clo[s], 0); l1 = np.maximum.accumulate(chi[s], 0)
            r0 = np.minimum.accumulate(clo[s][::-1], 0)[::-1]; r1 = np.maximum.accumulate(chi[s][::-1], 0)[::-1]
            ws = np.cumsum(w[s]); n = len(s); k = np.arange(1, n)
            cost = area(l0[k-1], l1[k-1]) * ws[k-1] + area(r0[k], r1[k]) * (ws[-1] - ws[k-1])
            j = np.argmin(cost)
            if cost[j] < best[0]: best = (cost[j], s, j + 1)
        _, s, k = best
        a = build(list(s[:k])); b = build(list(s[k:]))
        Ln.append(a); Rn.append(b)
        return len(Ln) - 1
    newroot = build(list(range(len(cut))))
    # order: children before parents: old nodes (order) then new ones in creation order
    used_order = list(order) + list(range(m, len(Ln)))
    return np.array(Ln), np.array(Rn), used_order, newroot, len(cut)

which = sys.argv[1]; N = int(sys.argv[2])
lo, hi = soup(N) if which == "soup" else terrain(int((N/2)**0.5*1.414), int((N/2)**0.5/1.414))
pad = 1e-5 * np.linalg.norm(hi.max(0) - lo.min(0)); lo, hi = lo - pad, hi + pad
keys = morton(lo, hi); o = np.argsort(keys, kind='stable'); lo, hi, keys = lo[o], hi[o], keys[o]
sys.setrecursionlimit(100000)
L, R, order, root = lbvh(keys); base = evaluate(L, R, order, root, lo, hi, "LBVH")
Lp, Rp, orderp, rootp, it = ploc(lo, hi, 16)
v = evaluate(Lp, Rp, orderp, rootp, lo, hi, "PLOC r=16"); print(f"   -> {100*(v/base-1):+.1f} %")
for T in (16, 64, 256, 1024):
    for name, (a, b, c, d) in (("PLOC", (Lp, Rp, orderp, rootp)), ("LBVH", (L, R, order, root))):
        Ln, Rn, on, rn, k = sah_top(a, b, c, d, lo, hi, T)
        # evaluate needs unused old nodes excluded: evaluate() walks `order` computing all; fine (unused nodes just computed)
        v = evaluate(Ln, Rn, on, rn, lo, hi, f"{name} subtrees <= {T} + SAH top ({k} clusters)"); print(f"   -> {100*(v/base-1):+.1f} %")
