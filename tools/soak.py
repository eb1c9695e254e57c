#!/usr/bin/env python3
"""Determinism soak: renders the same frames many times and checks that every repetition gives the same bits (a race in
the LDS pool / merge code would show up as a differing frame sooner or later).  Usage: tools/soak.py [repetitions]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optix_raytracer_amd import _lib as L, api, pathtracer as PT, scenes

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def soak(name, scene, w, h, spp, max_depth, shard_world=0):
    st = PT.make_state(scene, w, h, spp, 0, max_depth=max_depth)
    ob = PT.OutputBuffer(w, h, st.device)
    stream = torch.cuda.current_stream()
    if shard_world:
        ns = api.shard_num_samples(w, h, shard_world)
        compact = torch.zeros((ns, 4), dtype=torch.float32, device=st.device)
    seen = {}
    for r in range(reps):
        if shard_world:
            compact.zero_()
            st.context.launch_shard(st.params, w, h, r % shard_world, shard_world, compact.data_ptr(), stream=stream.cuda_stream)
            torch.cuda.synchronize()
            key, buf = r % shard_world, compact
        else:
            with torch.cuda.stream(st.stream):           # on the launch's stream: a launch writes pixels as soon as their paths end
                st.accum.zero_()
            PT.launchSubframe(ob, st)
            key, buf = 0, st.accum
        c = zlib.crc32(buf.cpu().numpy().tobytes())
        if seen.setdefault(key, c) != c:
            print("%s: repetition %d differs (crc %08x vs %08x)" % (name, r, c, seen[key]))
            sys.exit(1)
    print("%-56s %4d repetitions, identical (%s)" % (name, reps, ", ".join("%08x" % v for v in seen.values())), flush=True)
    PT.cleanupState(st)


soup = scenes.random_soup(1_000_000)
soak("headline frame (tiled closest, static any)", soup, 1920, 1080, 1, 1)
soak("soup 1M 960x540 depth 1 (path kernel from the start)", soup, 960, 540, 1, 1)
soak("soup 1M 512x512 unbounded (path kernel, long tails)", soup, 512, 512, 1, 0)
soak("soup 1M 1080p depth 8 (wavefront, then path kernel)", soup, 1920, 1080, 1, 8)
soak("shards of the headline frame, world 8", soup, 1920, 1080, 1, 1, shard_world=8)
soak("Cornell 512x512 unbounded", scenes.cornell_box(), 512, 512, 1, 0)
soak("soup 1M 256x256 spp 4 depth 4 (two rays per pool pass)", soup, 256, 256, 4, 4)


def soak_meshviewer(name, path, w, h, two_level, n):
    """the Whitted programs (MeshViewer mirror) with any-hit programs: subframe 0 again and again"""
    from optix_raytracer_amd import meshviewer as MV
    st = MV.MeshViewerState()
    MV.loadScene(st, path, two_level=two_level)
    MV.initCameraState(st)
    MV.initLaunchParams(st, w, h)
    out = MV.OutputBuffer(w, h, st.device)
    seen = None
    for r in range(n):
        st.params.subframe_index = 0
        MV.updateState(st)
        MV.launchSubframe(out, st)
        c = zlib.crc32(st.accum.cpu().numpy().tobytes())
        if seen is None:
            seen = c
        elif seen != c:
            print("%s: repetition %d differs (crc %08x vs %08x)" % (name, r, c, seen))
            sys.exit(1)
    print("%-56s %4d repetitions, identical (%08x)" % (name, n, seen), flush=True)
    MV.cleanup(st)


import pathlib, tempfile
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _png_fixture as PF
rng = np.random.default_rng(11)
film = np.zeros((8, 8, 4), np.int64); film[..., :3] = rng.integers(60, 255, (8, 8, 3)); film[..., 3] = rng.integers(40, 230, (8, 8))
fence = np.zeros((8, 8, 4), np.int64); fence[..., :3] = 200; fence[..., 3] = np.where((np.add.outer(np.arange(8), np.arange(8)) % 3) == 0, 0, 255)
alpha = scenes.write_instanced_gltf(pathlib.Path(tempfile.mkdtemp()) / "alpha.gltf", n_instances=300, seed=3,
                                    alpha_textures=(PF.encode(film, 6, 8)[0], PF.encode(fence, 6, 8)[0]))
soak_meshviewer("MeshViewer, 300 alpha instances, one level (any-hit programs)", alpha, 1280, 720, False, max(reps // 3, 20))
soak_meshviewer("MeshViewer, 300 alpha instances, two-level (any-hit, seeded)", alpha, 1280, 720, True, max(reps // 3, 20))
