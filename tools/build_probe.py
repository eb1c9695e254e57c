#!/usr/bin/env python3
"""Build time and traversal speed of the two BVH builders on the headline scene (MIPT_OPT_BUILD 0 host SAH, 1 device LBVH)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optix_raytracer_amd import _lib as L, pathtracer as PT, scenes
W, H = 1920, 1080
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scene = scenes.random_soup(n)
for mode in (0, 1, 1):
    st = PT.PathTracerState()
    PT.createContext(st, 0)
    st.context.set_option(L.OPT_BUILD, mode)
    t0 = time.perf_counter()
    PT.buildMeshAccel(st, scene)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    PT.createSBT(st); PT.initLaunchParams(st, W, H, 1)
    cam = scene["camera"]
    PT.handleCameraUpdate(st, PT.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], 1.0))
    st.context.set_option(L.OPT_MAX_DEPTH, 1)
    ob = PT.OutputBuffer(W, H, st.device)
    for _ in range(5):
        PT.launchSubframe(ob, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        PT.launchSubframe(ob, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    info = st.context.accel_info(st.gas_handle)
    print("builder %d: build %.4f s (accel_build call incl. input upload %.3f s), %d nodes, depth %d; frame %.3f ms" % (
        mode, info["build_seconds"], wall, info["num_nodes"], info["depth"], ms), flush=True)
    # refit to slightly moved vertices (mipt_accel_update), then the frame again
    import numpy as np
    rng = np.random.default_rng(3)
    v = np.array(scene["vertices"], np.float32, copy=True)
    v[:, :3] += rng.normal(0.0, 0.002, (v.shape[0], 3)).astype(np.float32)
    st.d_vertices.copy_(torch.from_numpy(v))
    d_mat = torch.from_numpy(np.ascontiguousarray(scene["mat_index"]).astype(np.int32)).cuda()
    inputs = [{"vertices": st.d_vertices.data_ptr(), "num_vertices": v.shape[0], "stride": 16, "sbt_index_offsets": d_mat.data_ptr()}]
    st.context.accel_update(st.gas_handle, inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        st.context.accel_update(st.gas_handle, inputs)
    torch.cuda.synchronize()
    upd = (time.perf_counter() - t0) / 5 * 1e3
    t0 = time.perf_counter()
    for _ in range(20):
        PT.launchSubframe(ob, st)
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / 20 * 1e3
    print("           refit %.3f ms; frame over the refitted tree %.3f ms" % (upd, ms2), flush=True)
    PT.cleanupState(st)
