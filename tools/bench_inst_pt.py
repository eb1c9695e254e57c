#!/usr/bin/env python3
"""The path tracer (optixPathTracer's programs) over a two-level structure: ms per launch with the two-level path kernel (k_tail_pool_inst,
round 3) against wavefront bounces to the end (MIPT_TAIL_PATHS=0: what a launch over an instance structure ran before), and against the
same scene flattened under one BVH."""
import os, pathlib, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from optix_raytracer_amd import api, scenes, _lib as L, pathtracer as PT

tmp = pathlib.Path(tempfile.mkdtemp())
path = scenes.write_instanced_gltf(tmp / "inst.gltf", n_instances=150, seed=3)
gl = api.Scene(path)
flat = scenes.from_gltf(path)
meshes, insts = scenes.gltf_two_level(gl)
for w, h, depth in ((256, 256, 0), (512, 512, 0), (1920, 1080, 0), (1920, 1080, 4)):
    row = []
    for mode in ("two-level", "two-level, wavefront only", "flattened"):
        ctx = api.Context(0)
        if mode == "two-level, wavefront only":
            ctx.set_option(L.OPT_TAIL_PATHS, 0)
        keep = []
        if mode == "flattened":
            dv = torch.from_numpy(np.ascontiguousarray(flat["vertices"], np.float32)).cuda()
            dm = torch.from_numpy(flat["mat_index"].astype(np.int32)).cuda()
            handle = ctx.accel_build([{"vertices": dv.data_ptr(), "num_vertices": flat["vertices"].shape[0], "stride": 16, "sbt_index_offsets": dm.data_ptr()}])
        else:
            gas = []
            n_mat = flat["diffuse"].shape[0]
            for verts, mat in meshes:
                dv = torch.from_numpy(np.ascontiguousarray(verts)).cuda()
                dm = torch.from_numpy(np.where(mat == 0xFFFFFFFF, n_mat - 1, mat).astype(np.int32)).cuda()
                keep += [dv, dm]
                gas.append(ctx.accel_build([{"vertices": dv.data_ptr(), "num_vertices": verts.shape[0], "stride": 16, "sbt_index_offsets": dm.data_ptr()}]))
            handle = ctx.accel_build_instances([(gas[m], xf) for m, xf in insts])
        st = PT.PathTracerState()
        st.device = torch.device("cuda", 0); st.context = ctx; st.scene = flat
        st.d_vertices = torch.from_numpy(np.ascontiguousarray(flat["vertices"], np.float32)).cuda()
        st.gas_handle = handle
        PT.createSBT(st); PT.initLaunchParams(st, w, h, 1)
        cam = flat["camera"]
        PT.handleCameraUpdate(st, PT.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], 1.0))
        ctx.set_option(L.OPT_MAX_DEPTH, depth)
        ob = PT.OutputBuffer(w, h, st.device)
        for _ in range(3):
            PT.launchSubframe(ob, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            PT.launchSubframe(ob, st)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        s = ctx.launch_stats()
        row.append("%s %.3f ms" % (mode, ms))
        ctx.destroy()
    print("%4dx%-4d depth %s (%d closest + %d shadow rays): %s" % (w, h, depth or "RR", s["closest_rays"], s["shadow_rays"], " | ".join(row)), flush=True)
