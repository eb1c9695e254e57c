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


KAT = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "ref_kat_imgui.json")


def _fb(bits3):
    return np.array(bits3, np.uint32).view(np.float32)


def test_layouts_equal_the_reference_classes():
    """sizeof / offsetof of the reference's own classes, printed by oracle/ref_kat_imgui.cpp from SDK/imgui_test/*.h: the
    product's C structs (include/mipt.h via _lib.py) and the oracle's have the same layouts."""
    import json
    lay = json.load(open(KAT))["layout"]
    cam_map = {"m_eye": "eye", "m_lookat": "lookat", "m_up": "up", "m_ortho": "ortho", "m_fov": "fov", "m_fd": "fd",
               "m_aperture": "aperture", "m_speed": "speed", "u": "u", "v": "v", "w": "w"}
    for T in (L.MiptImguiCamera, O.OrcImguiCamera):
        assert C.sizeof(T) == lay["Camera"]["size"]
        for ref, ours in cam_map.items():
            assert getattr(T, ours).offset == lay["Camera"][ref], ref
    for T in (L.MiptImguiLight, O.OrcImguiLight):
        assert C.sizeof(T) == lay["LightVariant"]["size"] and T.tag.offset == lay["LightVariant"]["_tag"] and T.a.offset == lay["LightVariant"]["_payload"]
    # the union members as MiptImguiLight documents them: a = first float3; b = next four floats; c = the rest
    assert lay["PointLight"] == {"size": 36, "m_position": 0, "m_lumi": 12, "m_dark": 24}
    assert lay["DirectionalLight"] == {"size": 40, "m_direction": 0, "m_lumi": 12, "m_jitter": 24, "m_dark": 28}
    assert lay["VolumetricLight"] == {"size": 40, "m_position": 0, "m_radius": 12, "m_lumi": 16, "m_dark": 28}
    assert lay["DiffuseMaterial"] == {"size": 12, "m_color": 0}
    P = L.MiptImguiParams
    assert C.sizeof(P) == lay["Params"]["size"]
    for k, off in lay["Params"].items():
        if k != "size":
            assert getattr(P, k).offset == off, k


def test_camera_known_answers_from_the_reference():
    """Camera::set_fd / compute_uvw / compute_ray of SDK/imgui_test/camera.h, run on the host by oracle/ref_kat_imgui.cpp (clang++:
    the two aperture draws are sibling arguments, evaluated left to right as nvcc's device compiler does)."""
    import json
    lib, plib = O.lib(), L.load()
    for e in json.load(open(KAT))["camera"]:
        for which in ("oracle", "product"):
            cam = (O.OrcImguiCamera if which == "oracle" else L.MiptImguiCamera)()
            cam.eye, cam.lookat, cam.up = (C.c_float * 3)(*_fb(e["eye"])), (C.c_float * 3)(*_fb(e["lookat"])), (C.c_float * 3)(*_fb(e["up"]))
            cam.ortho, cam.fov, cam.aperture, cam.speed = e["ortho"], _fb([e["fov"]])[0], _fb([e["aperture"]])[0], np.float32(0.01)
            if which == "oracle":
                lib.orc_imgui_camera_set_fd(C.byref(cam), _fb([e["fd_in"]])[0])
                lib.orc_imgui_camera_compute_uvw(C.byref(cam))
            else:
                plib.mipt_imgui_camera_set_fd(C.byref(cam), _fb([e["fd_in"]])[0])
                plib.mipt_imgui_camera_compute_uvw(C.byref(cam))
            assert np.float32(cam.fd).view(np.uint32) == e["m_fd"], which
            for k in ("u", "v", "w"):
                assert np.array(getattr(cam, k), np.float32).view(np.uint32).tolist() == e[k], (which, k)
            # the whole object, byte for byte (the three padding bytes behind `m_ortho` excepted)
            got = np.frombuffer(bytes(cam), np.uint32).copy()
            ref = np.array(e["object"], np.uint32)
            got[9] &= 0xff
            ref[9] &= 0xff
            assert np.array_equal(got, ref), which
        ocam = O.OrcImguiCamera.from_buffer_copy(bytes(cam))
        for r in e["rays"]:
            seed = C.c_uint32(r["seed0"])
            o, d = (C.c_float * 3)(), (C.c_float * 3)()
            lib.orc_imgui_compute_ray(C.byref(ocam), r["idx"][0], r["idx"][1], r["dim"][0], r["dim"][1], C.byref(seed), o, d)
            assert np.array(o, np.float32).view(np.uint32).tolist() == r["origin"] and np.array(d, np.float32).view(np.uint32).tolist() == r["direction"]
            lib.orc_imgui_compute_ray(C.byref(ocam), r["idx"][0], r["idx"][1], r["dim"][0], r["dim"][1], C.byref(seed), o, d)
            assert np.array(o, np.float32).view(np.uint32).tolist() == r["origin2"] and np.array(d, np.float32).view(np.uint32).tolist() == r["direction2"]
            assert seed.value == r["seed_after"]


def test_light_and_material_known_answers_from_the_reference():
    """LightVariant::wi / lumi (light.h:19-40) for point, directional (jittered) and volumetric lights and DiffuseMaterial::f:
    the reference object's 44 bytes are used AS the oracle's / the product's light struct, so the layout mapping is pinned too."""
    import json
    lib = O.lib()
    kat = json.load(open(KAT))
    for e in kat["lights"]:
        raw = np.array(e["object"], np.uint32).tobytes()
        assert len(raw) == 44
        ol = O.OrcImguiLight.from_buffer_copy(raw)
        pl = L.MiptImguiLight.from_buffer_copy(raw)
        assert ol.tag == pl.tag == e["tag"]
        for q in e["wi"]:
            seed = C.c_uint32(q["seed0"])
            wi, lumi = (C.c_float * 3)(), (C.c_float * 3)()
            p = (C.c_float * 3)(*_fb(q["p"]))
            lib.orc_imgui_light_wi(C.byref(ol), p, C.byref(seed), wi, lumi)
            assert np.array(wi, np.float32).view(np.uint32).tolist() == q["wi"]
            assert np.array(lumi, np.float32).view(np.uint32).tolist() == e["lumi"]
            lib.orc_imgui_light_wi(C.byref(ol), p, C.byref(seed), wi, lumi)
            assert np.array(wi, np.float32).view(np.uint32).tolist() == q["wi2"] and seed.value == q["seed_after"]
    # the host mirror builds the same bytes from (kind, parameters)
    from optix_raytracer_amd import imgui_test as IT2
    for e in kat["lights"]:
        w = np.array(e["object"], np.uint32).view(np.float32)
        if e["tag"] == 0:
            made = IT2.PointLight(w[0:3], w[3:6])
        elif e["tag"] == 1:
            made = IT2.DirectionalLight(w[0:3], w[3:6], w[6])
        else:
            made = IT2.VolumetricLight(w[0:3], w[3], w[4:7])
        assert bytes(made) == np.array(e["object"], np.uint32).tobytes()
    for e in kat["diffuse"]:
        assert e["f"] == e["color"] == e["object"] and e["seed_after"] == 5          # f() returns the colour, draws nothing


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
