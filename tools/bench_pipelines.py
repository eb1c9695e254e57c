#!/usr/bin/env python3
"""Times the two other callers of the traversal kernels on one GPU: the optixMeshViewer mirror (whitted.cu programs over a
generated glTF scene, opaque and with textured / alpha materials) and the imgui_test programs (NBT mesh, 4 lights)."""
import os, sys, time, pathlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from optix_raytracer_amd import api, scenes, meshviewer as MV, imgui_test as IT
from optix_raytracer_amd.pathtracer import OutputBuffer

tmp = pathlib.Path(tempfile.mkdtemp())


def meshviewer(name, path, w, h, reps, two_level=False, inst_form=None):
    """two_level: per-mesh structures + the instance structure (what the reference and the C++ sutil::Scene build); inst_form = 2 pins the
    one-ray-per-lane walk of the structure (MIPT_INST_FORM: the form a structure of ONE instance ran until round 4)"""
    if inst_form is not None:
        os.environ["MIPT_INST_FORM"] = str(inst_form)
    else:
        os.environ.pop("MIPT_INST_FORM", None)
    st = MV.MeshViewerState()
    MV.loadScene(st, str(path), two_level=two_level)
    MV.initCameraState(st)
    MV.initLaunchParams(st, w, h)
    out = OutputBuffer(w, h, st.device)
    MV.updateState(st)
    MV.launchSubframe(out, st)
    s = st.context.launch_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        st.params.subframe_index += 1
        MV.launchSubframe(out, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    rays = s["closest_rays"] + s["shadow_rays"]
    print("%-62s %8.3f ms/subframe %9.1f Mray/s  (%d radiance + %d occlusion rays, %d levels)" % (
        name, ms, rays / ms / 1e3, s["closest_rays"], s["shadow_rays"], s["bounces"]), flush=True)
    MV.cleanup(st)


def imgui(name, n_quads, w, h, spf, reps):
    rng = np.random.default_rng(7)
    # a bumpy height field of n_quads x n_quads cells with smooth normals, three packed float3 per triangle
    g = n_quads + 1
    xs, zs = np.meshgrid(np.linspace(-4, 4, g), np.linspace(-4, 4, g), indexing="ij")
    ys = 0.3 * np.sin(xs * 2.1) * np.cos(zs * 1.7) + 0.02 * rng.standard_normal(xs.shape)
    P = np.stack([xs, ys, zs], -1).astype(np.float32)
    gy, gx = np.gradient(ys, 8.0 / n_quads)
    N = np.stack([-gy, np.ones_like(ys), -gx], -1)
    N = (N / np.linalg.norm(N, axis=-1, keepdims=True)).astype(np.float32)
    i, j = np.meshgrid(np.arange(n_quads), np.arange(n_quads), indexing="ij")
    quad = lambda A: np.stack([A[i, j], A[i + 1, j], A[i + 1, j + 1], A[i, j], A[i + 1, j + 1], A[i, j + 1]], 2).reshape(-1, 3)
    v, n = quad(P), quad(N)
    ctx = api.Context(0)
    dev = torch.device("cuda", 0)
    mat = (np.arange(v.shape[0] // 3) % 3).astype(np.int32)
    gas = IT.TriangleGAS(ctx, dev, v, n, mat)
    cam, lights, mats = IT.default_scene_objects()
    cam.set_eye((0.5, 3.5, -9.0)); cam.set_lookat((0.0, 0.0, 0.0)); cam.set_fd(1.0); cam.compute_uvw()
    params = IT.Params(dev)
    params.p.image_width, params.p.image_height, params.p.samples_per_frame = w, h, spf
    params.set_camera(cam); params.set_lights(lights); params.set_materials(mats); params.set_mat_indices(gas.mat_indices)
    params.p.normals = gas.d_normals.data_ptr(); params.p.vertices = gas.d_vertices.data_ptr(); params.p.handle = gas.handle
    params.alloc_film()
    image = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    IT.render_frame(ctx, params, w, h, image, dirty=True)
    s = ctx.launch_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        IT.render_frame(ctx, params, w, h, image)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    rays = s["closest_rays"] + s["shadow_rays"]
    print("%-62s %8.3f ms/frame    %9.1f Mray/s  (%d radiance + %d occlusion rays, %d lights)" % (
        name, ms, rays / ms / 1e3, s["closest_rays"], s["shadow_rays"], len(lights)), flush=True)
    ctx.destroy()


meshviewer("meshviewer: generated glTF 4.2k tris, 1920x1080", scenes.write_rooms_gltf(tmp / "a.gltf", n_boxes=350), 1920, 1080, 10)
meshviewer("meshviewer: generated glTF 240k tris, 1920x1080", scenes.write_rooms_gltf(tmp / "b.gltf", n_boxes=20_000), 1920, 1080, 10)
meshviewer("meshviewer: generated glTF 240k tris, 3840x2160", tmp / "b.gltf", 3840, 2160, 5)
import _gltf_fixture as G


def _mk(name):
    d = tmp / name
    d.mkdir(exist_ok=True)
    return d


meshviewer("meshviewer: textured fixture (PNG/JPEG textures), 1920x1080", G.write_textured(_mk("t")), 1920, 1080, 10)
meshviewer("meshviewer: alpha MASK/BLEND fixture (any-hit programs), 1920x1080", G.write_alpha(_mk("al")), 1920, 1080, 10)
# structures of ONE instance (one mesh under one node: most glTF files), as sutil::Scene builds them: flattened / two-level / two-level on the
# one-ray-per-lane walk (what such a structure ran until round 4)
duck = os.path.join(ROOT, "tests", "golden", "duck", "Duck.gltf")
for label, path in (("Duck.gltf 4.2k tris", duck), ("generated glTF 240k tris", tmp / "b.gltf")):
    meshviewer("meshviewer: %s, 1920x1080, flattened" % label, path, 1920, 1080, 10)
    meshviewer("meshviewer: %s, 1920x1080, two-level" % label, path, 1920, 1080, 10, two_level=True)
    meshviewer("meshviewer: %s, 1920x1080, two-level, per-lane walk" % label, path, 1920, 1080, 10, two_level=True, inst_form=2)
imgui("imgui_test: 2x200x200-triangle height field, 1024x768 spf 1", 200, 1024, 768, 1, 10)
imgui("imgui_test: 2x700x700-triangle height field, 1920x1080 spf 4", 700, 1920, 1080, 4, 5)
