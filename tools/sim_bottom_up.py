#!/usr/bin/env python3
"""CPU model (no GPU): node visits / triangle tests of an occlusion ray that starts at a primary hit point, walked TOP-DOWN from the
root (today's any-hit traversal) against BOTTOM-UP from the leaf the hit lies in (visit the leaf's parent, then at every ancestor the
sibling subtrees only, up to the root: the whole tree is covered exactly once, so the boolean result is the same).  BVH2 over the
headline scene from the product's host builder (mipt_host_bvh_create); a sample of the headline frame's shadow rays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from optix_raytracer_amd import api, scenes

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
scene = scenes.random_soup(n_tris)
nodes, tris, info = api.host_bvh(scene["vertices"])
nodes = nodes.reshape(-1, 16)
ref = nodes[:, 12:14].copy().view(np.int32)
# child c box: c0x[2] c0y[2] | c1x[2] c1y[2] | c0z[2] c1z[2]
lo = np.stack([np.stack([nodes[:, 0], nodes[:, 2], nodes[:, 8]], 1), np.stack([nodes[:, 4], nodes[:, 6], nodes[:, 10]], 1)], 1)
hi = np.stack([np.stack([nodes[:, 1], nodes[:, 3], nodes[:, 9]], 1), np.stack([nodes[:, 5], nodes[:, 7], nodes[:, 11]], 1)], 1)
tris = tris.reshape(-1, 12)
nn = nodes.shape[0]
parent = np.full(nn, -1, np.int64)
for c in (0, 1):
    m = ref[:, c] >= 0
    parent[ref[m, c]] = np.flatnonzero(m)
leaf_parent = np.zeros(tris.shape[0], np.int64)          # per triangle (leaf order): the node that holds its leaf
for c in (0, 1):
    m = np.flatnonzero(ref[:, c] < 0)
    d = ~ref[m, c]
    for k in range(4):
        ok = (d & 15) > k
        leaf_parent[(d[ok] >> 4) + k] = m[ok]
print("BVH2: %d nodes, depth %d" % (nn, info["depth"]))


def slab(o, idir, tmin, tmax, l, h):
    a, b = (l - o) * idir, (h - o) * idir
    tn = max(np.minimum(a, b).max(), tmin); tf = min(np.maximum(a, b).min(), tmax)
    return tn <= tf, tn


def tri_hit(o, d, tmin, tmax, t):
    v0, e1, e2 = t[0:3], t[4:7], t[8:11]
    p = np.cross(d, e2); det = e1 @ p
    if det == 0: return None
    inv = 1.0 / det; tv = o - v0
    u = (tv @ p) * inv; q = np.cross(tv, e1); v = (d @ q) * inv; tt = (e2 @ q) * inv
    if u >= 0 and v >= 0 and u + v <= 1 and tmin < tt < tmax: return tt
    return None


def walk(o, d, tmin, tmax, start, closest, cnt, skip_child=-2):
    """DFS from node `start`; returns best t or None; counts into cnt = [nodes, tris]"""
    idir = 1.0 / np.where(np.abs(d) > 1e-20, d, 1e-20)
    best = None
    st = [start]
    while st:
        x = st.pop()
        if x >= 0:
            cnt[0] += 1
            h = []
            for c in (0, 1):
                if ref[x, c] == skip_child: continue
                ok, tn = slab(o, idir, tmin, tmax if best is None else best, lo[x, c], hi[x, c])
                if ok: h.append((tn, ref[x, c]))
            h.sort(key=lambda e: -e[0])
            for _, r in h: st.append(int(r))
        else:
            dsc = ~x
            for k in range(dsc & 15):
                cnt[1] += 1
                t = tri_hit(o, d, tmin, tmax, tris[(dsc >> 4) + k])
                if t is not None:
                    if not closest: return t
                    if best is None or t < best: best = t
    return best


rng = np.random.default_rng(1)
cam = scene["camera"]
eye = np.array(cam["eye"]); look = np.array(cam["lookat"])
L = scene["light"]
corner, v1, v2, nrm = (np.array(L[k], np.float64) for k in ("corner", "v1", "v2", "normal"))
td = bu = np.zeros(2); tdc = [0, 0]; buc = [0, 0]; n_sh = 0; occl = 0; up_levels = []
tri_of_prim = None
while n_sh < n_sample:
    # a camera ray through a random point of the cube's front face region
    tgt = np.array([rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0, 1)])
    d = tgt - eye; d /= np.linalg.norm(d)
    c0 = [0, 0]
    # closest hit top-down, remembering the leaf slot of the hit
    idir = 1.0 / np.where(np.abs(d) > 1e-20, d, 1e-20)
    best, best_k = None, -1
    st = [0]
    while st:
        x = st.pop()
        if x >= 0:
            h = []
            for c in (0, 1):
                ok, tn = slab(eye, idir, 0.01, 1e16 if best is None else best, lo[x, c], hi[x, c])
                if ok: h.append((tn, ref[x, c]))
            h.sort(key=lambda e: -e[0])
            for _, r in h: st.append(int(r))
        else:
            dsc = ~x
            for k in range(dsc & 15):
                t = tri_hit(eye, d, 0.01, 1e16, tris[(dsc >> 4) + k])
                if t is not None and (best is None or t < best): best, best_k = t, (dsc >> 4) + k
    if best is None: continue
    P = eye + best * d
    tr = tris[best_k]
    N = np.cross(tr[4:7], tr[8:11]); N /= np.linalg.norm(N)
    if N @ -d < 0: N = -N
    lp = corner + v1 * rng.uniform() + v2 * rng.uniform()
    Lv = lp - P; dist = np.linalg.norm(Lv); Ld = Lv / dist
    if not (N @ Ld > 0 and -(nrm @ Ld) > 0): continue
    n_sh += 1
    # top-down any hit
    r1 = walk(P, Ld, 0.01, dist - 0.01, 0, False, tdc)
    # bottom-up: the leaf's node first (all of it), then every ancestor's other child
    x = int(leaf_parent[best_k]); r2 = walk(P, Ld, 0.01, dist - 0.01, x, False, buc); lv = 0
    while r2 is None and parent[x] >= 0:
        p = int(parent[x]); lv += 1
        # the ancestor's OTHER child only: one box test (counted as a node visit), then its subtree
        c = 0 if ref[p, 0] != x else 1
        buc[0] += 1
        idir2 = 1.0 / np.where(np.abs(Ld) > 1e-20, Ld, 1e-20)
        ok, _ = slab(P, idir2, 0.01, dist - 0.01, lo[p, c], hi[p, c])
        if ok:
            r = int(ref[p, c])
            if r >= 0: r2 = walk(P, Ld, 0.01, dist - 0.01, r, False, buc)
            else:
                for k in range((~r) & 15):
                    buc[1] += 1
                    if tri_hit(P, Ld, 0.01, dist - 0.01, tris[((~r) >> 4) + k]) is not None: r2 = 1.0; break
        x = p
    assert (r1 is None) == (r2 is None)
    occl += r1 is not None; up_levels.append(lv)
print("%d shadow rays, %d occluded" % (n_sh, occl))
print("top-down : %.1f node visits + %.1f triangle tests per ray" % (tdc[0] / n_sh, tdc[1] / n_sh))
print("bottom-up: %.1f node visits + %.1f triangle tests per ray; levels climbed: mean %.1f, p90 %d, max %d" % (
    buc[0] / n_sh, buc[1] / n_sh, np.mean(up_levels), np.percentile(up_levels, 90), max(up_levels)))
