#!/usr/bin/env python3
"""What would opening instances into the top level (partial re-braiding) buy?  Emulation with the existing API: every mesh of the
instanced scene is cut into PARTS pieces along the Morton order of its triangles' centroids, each piece a mesh of its own, every
instance becomes PARTS instances with the same transform -- the structure a braided top level with PARTS entries per instance would
be.  Times mipt_trace_closest on the 1080p primary rays and mipt_trace_any on segments between random points of the scene box, over
the plain two-level structure, the cut one, and the flattened scene.
usage (GPU box): python tools/probe_braid.py [n_instances] [parts,parts,...]"""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from optix_raytracer_amd import _lib as L, api, scenes

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
parts_list = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 8, 16]
tmp = pathlib.Path(tempfile.mkdtemp())
path = scenes.write_instanced_gltf(tmp / "inst.gltf", n_instances=n_inst, seed=3)
gl = api.Scene(str(path))
flat = scenes.from_gltf(str(path))
meshes, insts = scenes.gltf_two_level(gl)
ctx = api.Context(0)
ctx.set_option(L.OPT_PROFILE, 1)


def morton_order(v):
    c = v[:, :3].reshape(-1, 3, 3).mean(1)
    lo, hi = c.min(0), c.max(0)
    q = np.clip(((c - lo) / np.maximum(hi - lo, 1e-20) * 1023).astype(np.int64), 0, 1023)
    code = np.zeros(len(c), np.int64)
    for b in range(10):
        for k in range(3):
            code |= ((q[:, k] >> b) & 1) << (3 * b + k)
    return np.argsort(code, kind="stable")


def build(parts):
    keep, gas_of = [], []
    for verts, mat in meshes:
        tri = verts.reshape(-1, 3, 4)
        order = morton_order(verts)
        k = max(1, min(parts, len(order) // 8))
        pieces = []
        for chunk in np.array_split(order, k):
            v = np.ascontiguousarray(tri[chunk].reshape(-1, 4))
            dv = torch.from_numpy(v).cuda()
            keep.append(dv)
            pieces.append(ctx.accel_build([{"vertices": dv.data_ptr(), "num_vertices": v.shape[0], "stride": 16}]))
        gas_of.append(pieces)
    inst = [(g, xf) for m, xf in insts for g in gas_of[m]]
    return ctx.accel_build_instances(inst), keep, len(inst)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    ctx.launch_timings()
    for _ in range(reps):
        fn()
    t = ctx.launch_timings()
    return (t["ms_trace_closest"] + t["ms_trace_any"]) / reps


import _oracle as O  # noqa: E402  (primary rays only)
osc = O.Scene(flat)
rays = osc.primary_rays(scenes.launch_desc(flat, 1920, 1080))
rng = np.random.default_rng(1)
n = rays.shape[0]
seg = np.zeros((n, 8), np.float32)
a = rng.uniform(-6, 6, (n, 3)) * (1, 0.35, 1) + (0, 1.2, 0)
b = rng.uniform(-6, 6, (n, 3)) * (1, 0.35, 1) + (0, 1.2, 0)
d = b - a
ln = np.linalg.norm(d, axis=1, keepdims=True)
seg[:, :3], seg[:, 4:7], seg[:, 3], seg[:, 7] = a, d / ln, 0.01, ln[:, 0]
d_rays, d_seg = torch.from_numpy(rays).cuda(), torch.from_numpy(seg).cuda()
d_hits = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
d_occ = torch.zeros(n, dtype=torch.int32, device="cuda")
dv = torch.from_numpy(np.ascontiguousarray(flat["vertices"], np.float32)).cuda()
one = ctx.accel_build([{"vertices": dv.data_ptr(), "num_vertices": flat["vertices"].shape[0], "stride": 16}])
print("%d instances, %d triangles; %d primary rays, %d segments" % (len(insts), flat["vertices"].shape[0] // 3, n, n))
for label, h in [("flattened", one)] + [("two-level, %2d part(s) per mesh" % p, None) for p in parts_list]:
    extra = ""
    if h is None:
        p = int(label.split(",")[1].split()[0])
        h, keep, k = build(p)
        extra = " (%d top-level instances)" % k
    c = timed(lambda: ctx.trace_closest(h, d_rays.data_ptr(), n, d_hits.data_ptr()))
    s = timed(lambda: ctx.trace_any(h, d_seg.data_ptr(), n, d_occ.data_ptr()))
    print("%-34s closest %7.3f ms   any %7.3f ms   (hits %d, occluded %d)%s" % (label, c, s, int((d_hits[:, 0] >= 0).sum()), int(d_occ.sum()), extra), flush=True)
