#!/usr/bin/env python3
"""optixMeshViewer's programs over an instanced glTF (scenes.write_instanced_gltf): the two-level (instance) structure that
sutil::Scene builds against the same scene flattened into one triangle BVH -- ms per 1080p subframe, same frames."""
import os
import pathlib
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from optix_raytracer_amd import meshviewer as MV, scenes

tmp = pathlib.Path(tempfile.mkdtemp())
counts = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [150, 2000]
for n_inst in counts:
    path = scenes.write_instanced_gltf(tmp / ("inst%d.gltf" % n_inst), n_instances=n_inst, seed=3)
    frames = {}
    for two_level in (True, False):
        st, img = MV.render(path, 1920, 1080, subframes=2, two_level=two_level)
        out = MV.OutputBuffer(1920, 1080, st.device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            MV.updateState(st)
            MV.launchSubframe(out, st)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        frames[two_level] = img
        info = st.context.accel_info(st.params.handle) if hasattr(st.params, "handle") else {}
        print("%5d instances, %-10s %8.3f ms per 1080p subframe   %s" % (n_inst, "two-level" if two_level else "flattened", ms,
              {k: info[k] for k in ("num_nodes", "num_triangles", "num_instances", "device_bytes") if k in info}), flush=True)
        MV.cleanup(st)
    print("      pixels that differ between the two: %d of %d" % (int((frames[True] != frames[False]).any(-1).sum()), 1920 * 1080))
