#!/usr/bin/env python3
"""The Whitted any-hit programs (MASK / BLEND materials) on a LARGE scene: 2 000 instances of an alpha-textured blob and box (782 k
triangles), flattened and two-level, against the same scene with opaque materials -- what the one-ray-per-thread walk costs."""
import os, pathlib, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _png_fixture as PF
from optix_raytracer_amd import meshviewer as MV, scenes
tmp = pathlib.Path(tempfile.mkdtemp())
rng = np.random.default_rng(11)
film = np.zeros((8, 8, 4), np.int64); film[..., :3] = rng.integers(60, 255, (8, 8, 3)); film[..., 3] = rng.integers(40, 230, (8, 8))
fence = np.zeros((8, 8, 4), np.int64); fence[..., :3] = 200; fence[..., 3] = np.where((np.add.outer(np.arange(8), np.arange(8)) % 3) == 0, 0, 255)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
only = sys.argv[2] if len(sys.argv) > 2 else ""          # "alpha": the MASK / BLEND scene alone (for a profiler run)
for alpha in ((True, "MASK", "OPAQUE") if only == "alpha" else (False, True, "MASK", "OPAQUE")):
    path = scenes.write_instanced_gltf(tmp / ("a%s.gltf" % alpha), n_instances=n, seed=3, blob_mode=alpha if isinstance(alpha, str) else "BLEND",
                                       alpha_textures=(PF.encode(film, 6, 8)[0], PF.encode(fence, 6, 8)[0]) if alpha else None)
    for two_level in (False, True):
        st, img = MV.render(path, 1920, 1080, subframes=2, two_level=two_level)
        out = MV.OutputBuffer(1920, 1080, st.device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 5
        for _ in range(K):
            MV.updateState(st); MV.launchSubframe(out, st)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        s = st.context.launch_stats()          # (MIPT_PROBE_PRINT=1 with a MIPT_WH_AH_STATS probe library: the phase statistics on stderr)
        print("%d instances, %s materials, %-10s %8.3f ms per 1080p subframe (%d radiance + %d occlusion rays)" % (
            n, {False: "opaque    ", True: "BLEND+MASK", "MASK": "MASK      ", "OPAQUE": "boxes MASK"}[alpha], "two-level" if two_level else "flattened", ms, s["closest_rays"], s["shadow_rays"]), flush=True)
        MV.cleanup(st)
