#!/usr/bin/env python3
"""CPU study: lane utilisation of wave64 traversal schedules on the headline scene's primary rays
(uses the oracle's per-ray step profiles; no GPU)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
from optix_raytracer_amd import api, scenes  # noqa: E402

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
sc = scenes.random_soup(n_tris)
nodes, tris, info = api.host_bvh(sc["vertices"])
s = O.Scene(sc).set_bvh(nodes, tris)
W, H = 1920, 1080
launch = scenes.launch_desc(sc, W, H)
L = O.lib()
L.orc_trace_profile.restype = C.c_uint32
L.orc_trace_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
rng = np.random.default_rng(0)
C_IN, C_TRI = 50, 65     # wave instructions per inner step / triangle test

def profiles_for_tile(tx, ty):
    out = []
    rows = s.primary_rays(launch, rows=(ty * 8, ty * 8 + 8)).reshape(8, W, 8)
    buf = np.zeros(2 * 512, np.uint16)
    for y in range(8):
        for x in range(8):
            r = np.ascontiguousarray(rows[y, tx * 8 + x])
            n = L.orc_trace_profile(s._h, r.ctypes.data, 0, buf.ctypes.data, 512)
            out.append(buf[:2 * min(n, 512)].reshape(-1, 2).copy())
    return out

tiles = [(int(rng.integers(60, 180)), int(rng.integers(5, 130))) for _ in range(150)]
tot = {"useful": 0, "ww": 0, "uni": 0}
steps_all = []
for tx, ty in tiles:
    prof = profiles_for_tile(tx, ty)
    useful = sum(int(p[:, 0].sum()) * C_IN + int(p[p[:, 1] != 0xffff, 1].sum()) * C_TRI for p in prof)
    steps_all += [int(p[:, 0].sum()) for p in prof]
    # while-while: round r = max inner steps of round r, then max tris of round r
    R = max(len(p) for p in prof)
    ww = 0
    for r in range(R):
        inn = max((int(p[r, 0]) for p in prof if len(p) > r), default=0)
        tri = max((int(p[r, 1]) for p in prof if len(p) > r and p[r, 1] != 0xffff), default=0)
        ww += inn * C_IN + tri * C_TRI
    # unified: every lane advances one op per iteration; iteration cost = C_IN if any lane inner + C_TRI if any at leaf
    seqs = []
    for p in prof:
        q = []
        for a, b in p:
            q += [0] * int(a)
            if b != 0xffff:
                q += [1] * int(b)
        seqs.append(q)
    T = max(len(q) for q in seqs)
    uni = 0
    for t in range(T):
        ops = [q[t] for q in seqs if len(q) > t]
        uni += (C_IN if 0 in ops else 0) + (C_TRI if 1 in ops else 0)
    tot["useful"] += useful
    tot["ww"] += ww * 64
    tot["uni"] += uni * 64
print("tris", n_tris, "bvh", info)
print("mean inner steps/ray %.1f  max %d" % (np.mean(steps_all), max(steps_all)))
print("utilisation while-while (no refill): %.3f" % (tot["useful"] / tot["ww"]))
print("utilisation unified     (no refill): %.3f" % (tot["useful"] / tot["uni"]))

# ---- with ideal refill: a wave streams through many rays; a lane takes the next ray as soon as it retires
all_prof = []
for tx, ty in tiles[:60]:
    all_prof += profiles_for_tile(tx, ty)
def stream_ww(all_prof):
    it = iter(all_prof)
    lanes = [list(next(it)) for _ in range(64)]
    pos = [0] * 64
    cost = useful = 0
    exhausted = False
    while not exhausted:
        inn = tri = 0
        for l in range(64):
            a, b = lanes[l][pos[l]]
            a = int(a); b = 0 if b == 0xffff else int(b)
            inn = max(inn, a); tri = max(tri, b)
            useful += a * C_IN + b * C_TRI
            pos[l] += 1
            if pos[l] >= len(lanes[l]):
                try:
                    lanes[l] = list(next(it)); pos[l] = 0
                except StopIteration:
                    exhausted = True
                    pos[l] = 0
        cost += (inn * C_IN + tri * C_TRI) * 64
    return useful / cost
def stream_uni(all_prof, leaf_min=1):
    it = iter(all_prof)
    def seq(p):
        q = []
        for a, b in p:
            q += [0] * int(a)
            if b != 0xffff:
                q += [1] * int(b)
        return q if q else [0]
    lanes = [seq(next(it)) for _ in range(64)]
    pos = [0] * 64
    cost = useful = 0
    while True:
        ops = [lanes[l][pos[l]] for l in range(64)]
        n_in = ops.count(0); n_leaf = 64 - n_in
        do_leaf = n_leaf > 0 and (n_leaf >= leaf_min or n_in == 0)
        c = (C_IN if n_in else 0) + (C_TRI if do_leaf else 0)
        cost += c * 64
        done = False
        for l in range(64):
            if ops[l] == 0 or do_leaf:
                useful += C_IN if ops[l] == 0 else C_TRI
                pos[l] += 1
                if pos[l] >= len(lanes[l]):
                    try:
                        lanes[l] = seq(next(it)); pos[l] = 0
                    except StopIteration:
                        done = True; pos[l] = 0
        if done:
            return useful / cost
print("utilisation while-while + ideal refill: %.3f" % stream_ww(all_prof))
for lm in (1, 8, 16, 24, 32):
    print("utilisation unified + ideal refill, leaf_min=%2d: %.3f" % (lm, stream_uni(all_prof, lm)))
