#!/usr/bin/env python3
"""Determinism of the device builders: the same input built N times (default 12) must give the same bytes -- nodes and triangle records -- every
time, whatever way the atomics of the level loop fell.  usage: soak_builder.py [n_builds]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MIPT_BUILD_HOST_BELOW"] = "0"
import numpy as np
import torch
from optix_raytracer_amd import _lib as L, api, scenes
n_builds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
for name, sc in (("soup 1M", scenes.random_soup(1_000_000)), ("dense soup 200k", scenes.random_soup(200_000, edge_scale=0.05)), ("soup 3k", scenes.random_soup(3000))):
    v = np.ascontiguousarray(sc["vertices"], np.float32)
    dv = torch.from_numpy(v).cuda()
    inp = [{"vertices": dv.data_ptr(), "num_vertices": v.shape[0], "stride": 16}]
    for mode in (2, 4):
        ctx = api.Context(0)
        ctx.set_option(L.OPT_BUILD, mode)
        seen = {}
        for k in range(n_builds):
            h = ctx.accel_build(inp)
            nodes, tris = ctx.accel_export(h)
            d = hashlib.sha1(nodes.tobytes() + tris.tobytes()).hexdigest()[:12]
            seen[d] = seen.get(d, 0) + 1
            info = ctx.accel_info(h)
            ctx.accel_destroy(h)
        bad += len(seen) != 1
        print("%-16s builder %d: %d builds, %d records, %d nodes: %s" % (name, mode, n_builds, info["num_triangles"], info["num_nodes"],
              "identical (%s)" % list(seen)[0] if len(seen) == 1 else "DIFFERENT TREES %s" % seen), flush=True)
        ctx.destroy()
print("failures:", bad)
