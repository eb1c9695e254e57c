"""SDK/imgui_test path: host objects, NBT reader, and (GPU) the wavefront version of its programs against the
oracle's per-pixel restatement of imgui_test/optixTriangle.cu:103-268."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from optix_raytracer_amd import _lib as L
from optix_raytracer_amd import imgui_test as IT


def _mesh(n=400, seed=3):
    """a bumpy floor + random triangles with smooth-ish normals (3 per triangle, not normalised on purpose)"""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-3, 3, (n, 1, 3)) * (1, 0.15, 1) + (0, 0.3, 0)
    v = (c + rng.uniform(-0.6, 0.6, (n, 3, 3))).astype(np.float32)
    floor = np.array([[[-6, 0, -6], [-6, 0, 6], [6, 0, 6]], [[-6, 0, -6], [6, 0, 6], [6, 0, -6]]], np.float32)
    v = np.concatenate([floor, v]).reshape(-1, 3)
    fn = np.cross(v[1::3] - v[0::3], v[2::3] - v[0::3])
    fn = fn / np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-6)
    fn[fn[:, 1] < 0] *= -1
    nrm = (np.repeat(fn, 3, axis=0) + rng.normal(scale=0.2, size=v.shape)).astype(np.float32)
    mat = rng.integers(0, 28, v.shape[0]).astype(np.int32)
    return v, nrm, mat


def test_layouts_and_camera_helpers():
    assert C.sizeof(L.MiptImguiCamera) == 92 and C.sizeof(L.MiptImguiLight) == 44 and C.sizeof(L.MiptImguiParams) == 128
    P = L.MiptImguiParams
    for k, off in {"camera": 16, "dt": 24, "dirty": 28, "image": 32, "film": 40, "handle": 56, "normals": 64, "lights": 96,
                   "nlights": 104, "materials": 112, "nmaterials": 120}.items():
        assert getattr(P, k).offset == off, k
    cam, lights, mats = IT.default_scene_objects()
    oc = O.OrcImguiCamera.from_buffer_copy(cam.pack())
    oc.fd = 0.0
    O.lib().orc_imgui_camera_set_fd(C.byref(oc), 1.0)
    O.lib().orc_imgui_camera_compute_uvw(C.byref(oc))
    assert bytes(oc) == cam.pack()                       # product host helpers == oracle, bit for bit
    assert abs(np.linalg.norm(np.array(cam.c.w)) - 1.0) < 1e-5 and len(lights) == 4 and len(mats) == 28     # set_fd(1) scales w to unit length


def test_nbt_round_trip(tmp_path):
    v, n, _ = _mesh(50)
    for compress in (True, False):
        p = tmp_path / ("m%d.nbt" % compress)
        IT.write_nbt(p, {"a": (v[:60], n[:60]), "b": (v[60:], n[60:])}, compress)
        v2, n2, m2 = IT.load_nbt(p)
        assert np.array_equal(v2, v) and np.array_equal(n2, n) and m2.shape[0] == v.shape[0] and (m2 == 0).all()
    with pytest.raises(RuntimeError):
        IT.load_nbt(tmp_path / "missing.nbt")


def _oracle_frame(v, n, mat, nodes, tris, cam, lights, mats, w, h, spf, frames):
    verts4 = np.zeros((v.shape[0], 4), np.float32)
    verts4[:, :3] = v
    sc = {"vertices": verts4, "mat_index": np.zeros(v.shape[0] // 3, np.uint32), "emission": np.zeros((1, 3), np.float32),
          "diffuse": np.ones((1, 3), np.float32)}
    osc = O.Scene(sc).set_bvh(nodes, tris)
    ocam = O.OrcImguiCamera.from_buffer_copy(cam.pack())
    olights = (O.OrcImguiLight * max(len(lights), 1))()
    for i, l in enumerate(lights):
        olights[i] = O.OrcImguiLight.from_buffer_copy(bytes(l))
    mcol = np.ascontiguousarray(mats, np.float32)
    nrm = np.ascontiguousarray(n, np.float32)
    mi = np.ascontiguousarray(mat, np.int32)
    film = np.zeros((h, w, 3), np.float32)
    image = np.zeros((h, w, 4), np.uint8)
    st = O.OrcStats()
    dt = 0
    for fr in range(frames):
        dt += spf
        f = O.OrcImguiFrame(w, h, spf, dt, 1 if fr == 0 else 0, C.pointer(ocam), olights, len(lights),
                            mcol.ctypes.data_as(C.POINTER(C.c_float)), mcol.shape[0], mi.ctypes.data_as(C.POINTER(C.c_int32)),
                            nrm.ctypes.data_as(C.POINTER(C.c_float)), film.ctypes.data_as(C.POINTER(C.c_float)),
                            image.ctypes.data_as(C.POINTER(C.c_uint8)))
        rc = O.lib().orc_imgui_render_rows(osc._h, O.MODE_BVH, C.byref(f), 0, h, C.byref(st))
        assert rc == 0
    return film, image, st.as_dict()


@pytest.mark.gpu
@pytest.mark.parametrize("ortho,aperture,w,h,spf,frames", [(False, 0.0, 120, 90, 2, 1), (False, 0.05, 100, 75, 3, 2), (True, 0.0, 64, 64, 1, 3)])
def test_imgui_programs_match_the_oracle(tmp_path, ortho, aperture, w, h, spf, frames):
    import torch
    assert torch.cuda.is_available()
    from optix_raytracer_amd import api
    v, n, mat = _mesh()
    p = tmp_path / "mesh.nbt"
    IT.write_nbt(p, {"mesh": (v, n)})
    ctx = api.Context(0)
    dev = torch.device("cuda", 0)
    v2, n2, _ = IT.load_nbt(p)
    gas = IT.TriangleGAS(ctx, dev, v2, n2, mat[: v.shape[0] // 3])
    cam, lights, mats = IT.default_scene_objects()
    lights.append(IT.PointLight((-2.0, 3.0, -1.0), (0.3, 0.3, 0.25)))
    cam.set_eye((0.5, 2.5, -9.0)); cam.set_lookat((0.0, 0.2, 0.0)); cam.set_ortho(ortho); cam.set_aperture(aperture)
    cam.set_fd(1.0); cam.compute_uvw()
    params = IT.Params(dev)
    params.p.image_width, params.p.image_height, params.p.samples_per_frame = w, h, spf
    params.set_camera(cam); params.set_lights(lights); params.set_materials(mats); params.set_mat_indices(gas.mat_indices)
    params.p.normals = gas.d_normals.data_ptr(); params.p.vertices = gas.d_vertices.data_ptr(); params.p.handle = gas.handle
    params.alloc_film()
    image = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    tot_c = tot_s = 0
    for fr in range(frames):
        IT.render_frame(ctx, params, w, h, image, dirty=(fr == 0))
        s = ctx.launch_stats()
        tot_c += s["closest_rays"]; tot_s += s["shadow_rays"]
    nodes, tris = ctx.accel_export(gas.handle)
    film, img, ost = _oracle_frame(v, n, gas.mat_indices, nodes, tris, cam, lights, mats, w, h, spf, frames)
    assert np.array_equal(params.film.cpu().numpy().view(np.uint32), film.view(np.uint32))
    assert np.array_equal(image.cpu().numpy(), img)
    assert tot_c == ost["closest_rays"] == w * h * spf * frames and tot_s == ost["shadow_rays"] > 0
    assert (film.sum(2) > 0).mean() > 0.9 and np.isfinite(film).all()
    ctx.destroy()
