#!/usr/bin/env python3
"""Build, refit and frame time on the headline scene for both BVH builders (MIPT_OPT_BUILD 0 host SAH, 1 device LBVH) and every node
format (MIPT_OPT_NODE_FORMAT 0 fp32 nodes, 1 + packed fp16 BVH2 mirror, 2 + BVH4 mirror).  The refit is mipt_accel_update (=
OPTIX_BUILD_OPERATION_UPDATE, optixPathTracer.cpp:650-676) on vertices moved by a small random offset; it ends with the mirrors
remade on the device.  Kept per round under profiles/ (r03_build_probe.txt)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from optix_raytracer_amd import _lib as L, pathtracer as PT, scenes
W, H = 1920, 1080
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scene = scenes.random_soup(n)


def frames(st, ob, k=20):
    for _ in range(5):
        PT.launchSubframe(ob, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        PT.launchSubframe(ob, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


print("%d triangles, %dx%d, depth 1: builder (0 host SAH / 1 device LBVH / 2 device SAH / 3, 4 host, device SAH with spatial splits) x node format" % (n, W, H), flush=True)
only = [tuple(int(x) for x in c.split(":")) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else None     # e.g. "1:2,0:2"
for mode, fmt in [c for c in [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2), (2, 2), (4, 2), (3, 2)] if only is None or c in only]:
    st = PT.PathTracerState()
    PT.createContext(st, 0)
    st.context.set_option(L.OPT_BUILD, mode)
    st.context.set_option(L.OPT_NODE_FORMAT, fmt)
    t0 = time.perf_counter()
    PT.buildMeshAccel(st, scene)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    PT.createSBT(st); PT.initLaunchParams(st, W, H, 1)
    cam = scene["camera"]
    PT.handleCameraUpdate(st, PT.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], 1.0))
    st.context.set_option(L.OPT_MAX_DEPTH, 1)
    ob = PT.OutputBuffer(W, H, st.device)
    ms = frames(st, ob)
    info = st.context.accel_info(st.gas_handle)
    # refit to slightly moved vertices (mipt_accel_update), then the frame again
    rng = np.random.default_rng(3)
    v = np.array(scene["vertices"], np.float32, copy=True)
    v[:, :3] += rng.normal(0.0, 0.0005, (v.shape[0], 3)).astype(np.float32)
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
    ms2 = frames(st, ob)
    info2 = st.context.accel_info(st.gas_handle)
    print("builder %d format %d (in effect %d, %d mirror nodes, stack %d): build %8.2f ms (call incl. upload %7.1f ms), %d nodes depth %d over %d records, frame %.3f ms | "
          "refit %6.2f ms, frame %.3f ms (format in effect %d)" % (
              mode, fmt, info["node_format"], info["mirror_nodes"], info["stack_entries"], info["build_seconds"] * 1e3, wall * 1e3,
              info["num_nodes"], info["depth"], info["num_triangles"], ms, upd, ms2, info2["node_format"]), flush=True)
    PT.cleanupState(st)
