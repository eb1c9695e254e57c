"""The glTF / PBR programs of SDK/cuda/whitted.cu (opaque, untextured subset): oracle known answers on CPU,
HIP kernels against the oracle on the GPU, through the optixMeshViewer host mirror."""
import ctypes as C

import numpy as np
import pytest

import _gltf_fixture as FX
import _oracle as O

f32 = np.float32


def _quad(a, b, c, d):
    return [a, b, c, a, c, d]


def _plane_scene():
    """floor (material 0) at y = 0, a small blocker (material 1) at y = 1 above x,z in [-0.25, 0.25]"""
    v = _quad((-2, 0, -2), (-2, 0, 2), (2, 0, 2), (2, 0, -2)) + _quad((-.25, 1, -.25), (-.25, 1, .25), (.25, 1, .25), (.25, 1, -.25))
    verts = np.zeros((12, 4), f32)
    verts[:, :3] = np.asarray(v, f32)
    mat = np.asarray([0, 0, 1, 1], np.uint32)
    sc = O.Scene({"vertices": verts, "mat_index": mat, "emission": np.zeros((2, 3), f32), "diffuse": np.ones((2, 3), f32)})
    sc.build_bvh()
    return sc, verts, mat


MATS = [{"base_color": (0.8, 0.5, 0.2, 1.0), "metallic": 0.25, "roughness": 0.6, "emissive_factor": (0.01, 0.02, 0.03)},
        {"base_color": (0.1, 0.9, 0.1, 1.0), "metallic": 0.0, "roughness": 0.9, "emissive_factor": (0.0, 0.0, 0.0)}]


def _ggx_reference(P, N, eye, light, m):
    """double-precision restatement of whitted.cu:185-262 for one point light, unoccluded"""
    base = np.asarray(m["base_color"][:3], np.float64)
    F0 = 0.04
    diff_color = base * (1 - F0) * (1 - m["metallic"])
    spec_color = F0 + m["metallic"] * (base - F0)
    alpha = m["roughness"] ** 2
    L = np.asarray(light["position"], np.float64) - P
    L /= np.linalg.norm(L)
    V = eye - P
    V /= np.linalg.norm(V)
    H = (L + V) / np.linalg.norm(L + V)
    NdL, NdV, NdH, VdH = N @ L, N @ V, N @ H, V @ H
    F = spec_color + (1 - spec_color) * (1 - VdH) ** 5
    a2 = alpha * alpha
    G = 2 * NdL * NdV / (NdL * np.sqrt(NdV * NdV * (1 - a2) + a2) + NdV * np.sqrt(NdL * NdL * (1 - a2) + a2))
    D = a2 / (np.pi * (NdH * NdH * (a2 - 1) + 1) ** 2)
    return np.asarray(light["color"], np.float64) * light["intensity"] * NdL * ((1 - F) * diff_color / np.pi + F * G * D)


def _launch(w, h, eye, lookat, up=(0, 0, -1), fov=60.0, subframe=0):
    U, V, W = O.camera_uvw(eye, lookat, up, fov, w / h)
    return {"width": w, "height": h, "subframe": subframe, "eye": np.asarray(eye, f32), "U": U, "V": V, "W": W,
            "miss_color": (0.1, 0.2, 0.3)}


def test_oracle_whitted_known_answers():
    sc, verts, mat = _plane_scene()
    w = h = 65
    eye = np.asarray((0.0, 4.0, 0.0))
    la = _launch(w, h, eye, (0, 0, 0), fov=90.0)
    point = {"type": 0, "color": (1.0, 0.9, 0.8), "intensity": 4.0, "position": (0.0, 3.0, 0.0)}
    ambient = {"type": 1, "color": (0.05, 0.05, 0.1)}
    accum, frame, st = sc.render_whitted(la, [point, ambient], MATS, mat)
    assert st["closest_rays"] == w * h
    # corners leave the 4x4 floor at fov 90 from y = 4: miss colour, alpha 1
    assert np.allclose(accum[0, 0], (0.1, 0.2, 0.3, 1.0), rtol=0, atol=1e-7)
    # centre pixel hits the blocker (material 1), lit from straight above
    c = accum[h // 2, w // 2, :3]
    ref = _ggx_reference(np.array([0, 1.0, 0]), np.array([0, 1.0, 0]), eye, point, MATS[1]) + np.asarray(ambient["color"]) * MATS[1]["base_color"][:3]
    assert np.allclose(c, ref, rtol=2e-5)
    # a floor pixel right under the blocker's rim region is shadowed: emissive + ambient only
    # (light at y=3 over a blocker of half-width .25 at y=1 shadows |x|,|z| < .375 on the floor; the camera at y=4 sees
    #  the floor only outside |x| > 1/3 there, so x = 0.35 is visible and in shadow)
    px = int(round((0.35 / 4.0 + 1) * 0.5 * w - 0.5))
    shadowed = accum[h // 2, px, :3]
    amb = np.asarray(ambient["color"]) * MATS[0]["base_color"][:3] + MATS[0]["emissive_factor"]
    assert np.allclose(shadowed, amb, rtol=1e-6)
    # a floor pixel well outside the shadow: full formula
    px2 = int(round((1.5 / 4.0 + 1) * 0.5 * w - 0.5))
    x_world = ((px2 + 0.5) / w * 2 - 1) * 4.0
    lit = accum[h // 2, px2, :3]
    ref2 = _ggx_reference(np.array([x_world, 0, 0]), np.array([0, 1.0, 0]), eye, point, MATS[0]) + amb
    assert np.allclose(lit, ref2, rtol=5e-5)
    assert st["shadow_rays"] > 0 and np.isfinite(accum).all()
    assert frame[0, 0, 3] == 255


def test_oracle_whitted_progressive_lerp_and_brute_force():
    sc, verts, mat = _plane_scene()
    w, h = 40, 30
    lights = [{"type": 0, "color": (1, 1, 1), "intensity": 3.0, "position": (1.0, 2.5, 0.5)}]
    a0, _, _ = sc.render_whitted(_launch(w, h, (0.5, 3, 0.2), (0, 0, 0)), lights, MATS, mat)
    a0b, _, _ = sc.render_whitted(_launch(w, h, (0.5, 3, 0.2), (0, 0, 0)), lights, MATS, mat, mode=O.MODE_BRUTE)
    assert np.array_equal(a0.view(np.uint32), a0b.view(np.uint32))
    acc = a0.copy()
    a1, _, _ = sc.render_whitted(_launch(w, h, (0.5, 3, 0.2), (0, 0, 0), subframe=1), lights, MATS, mat, accum=acc)
    single, _, _ = sc.render_whitted(_launch(w, h, (0.5, 3, 0.2), (0, 0, 0), subframe=1), lights, MATS, mat,
                                     accum=np.zeros_like(a0))
    # lerp(prev, new, 1/2); `single` started from a zero accumulator so it holds new/2
    assert np.allclose(a1[..., :3], 0.5 * a0[..., :3] + single[..., :3], rtol=1e-6, atol=1e-7)


def test_shading_normals_flip_towards_the_viewer():
    sc, verts, mat = _plane_scene()
    w = h = 16
    lights = [{"type": 0, "color": (1, 1, 1), "intensity": 3.0, "position": (0.5, 2.5, 1.0)}]
    up_n = np.zeros((12, 4), f32); up_n[:, 1] = 1; up_n[:, 3] = 1
    dn_n = up_n.copy(); dn_n[:, 1] = -2.5                         # un-normalised and pointing away: must be flipped
    a, _, _ = sc.render_whitted(_launch(w, h, (0.0, 3, 0.0), (0, 0, 0)), lights, MATS, mat, normals4=up_n)
    b, _, _ = sc.render_whitted(_launch(w, h, (0.0, 3, 0.0), (0, 0, 0)), lights, MATS, mat, normals4=dn_n)
    c, _, _ = sc.render_whitted(_launch(w, h, (0.0, 3, 0.0), (0, 0, 0)), lights, MATS, mat, normals4=None)
    assert np.array_equal(a, b)
    assert np.allclose(a, c, rtol=1e-5, atol=1e-7)                # geometric normal of the planes is +-y as well


def test_scene_flatten_normals(tmp_path):
    from optix_raytracer_amd import api
    gltf, glb, expect = FX.write(tmp_path)
    for path in (gltf, glb):
        sc = api.Scene(path)
        n = sc.flatten_normals()
        assert n.shape == expect["normals"].shape
        assert np.array_equal(n[:, 3], expect["normals"][:, 3].astype(f32))
        assert np.allclose(n[:, :3], expect["normals"][:, :3], rtol=2e-5, atol=2e-6)
        tab = sc.material_table()
        assert len(tab) == sc.info()["num_materials"] + 1
        last = tab[len(tab) - 1]            # MaterialData() (cuda/MaterialData.h:44-51): what a primitive without a material gets (Scene.cpp:1423-1428)
        assert list(last.base_color) == [1.0] * 4 and last.roughness == 1.0 and last.metallic == 1.0 and last.base_color_tex.texture == -1


def test_meshviewer_default_lights_follow_the_sample():
    from optix_raytracer_amd import meshviewer as MV
    l0, l1 = MV.default_lights((-1.0, 0.0, -2.0), (3.0, 1.0, 2.0))     # extent (4,1,4): longest axis x, maxExtent 4
    assert np.allclose(l0["position"], (5.0, 4.5, 4.0)) and l0["intensity"] == 5.0 and l0["color"] == (1.0, 1.0, 0.8)
    assert np.allclose(l1["position"], (-3.0, 2.5, -2.0)) and l1["intensity"] == 3.0
    arr, n = MV.make_lights([l0, l1])
    assert n == 2 and C.sizeof(arr) == 72 and arr[1].type == MV.LIGHT_POINT and arr[0].falloff == MV.FALLOFF_QUADRATIC


def _oracle_for_state(st, nodes, tris):
    verts = st.d_vertices.cpu().numpy()
    mat = st.d_material_index.cpu().numpy().view(np.uint32)
    sc = O.Scene({"vertices": verts, "mat_index": np.zeros(mat.shape[0], np.uint32), "emission": np.zeros((1, 3), f32),
                  "diffuse": np.ones((1, 3), f32)})
    sc.set_bvh(nodes, tris)
    from optix_raytracer_amd import meshviewer as MV
    gf = MV.geometry_flags(st.materials)
    sc.set_prim_flags(gf[np.minimum(mat, len(gf) - 1)])
    mats = [{"base_color": list(m.base_color), "metallic": m.metallic, "roughness": m.roughness,
             "emissive_factor": list(m.emissive_factor), "alpha_mode": m.alpha_mode, "alpha_cutoff": m.alpha_cutoff,
             "double_sided": m.double_sided} for m in st.materials]
    lights = [{"type": l.type, "color": list(l.color), "intensity": l.intensity, "position": list(l.position),
               "falloff": l.falloff} for l in list(st.lights)[: st.params.lights.count]]
    normals = st.d_normals.cpu().numpy() if st.d_normals is not None else None
    return sc, mats, lights, mat, normals


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,subframes,custom", [(160, 120, 1, False), (97, 61, 3, True), (1024, 768, 1, True)])   # the last: tiled closest-hit kernel with back-face culling
def test_whitted_programs_match_the_oracle(tmp_path, w, h, subframes, custom):
    import torch
    assert torch.cuda.is_available()
    from optix_raytracer_amd import meshviewer as MV
    gltf, _, _ = FX.write(tmp_path)
    lights = None
    if custom:
        lights = [{"type": MV.LIGHT_AMBIENT, "color": (0.02, 0.03, 0.05)},
                  {"type": MV.LIGHT_POINT, "color": (1.0, 0.7, 0.4), "intensity": 6.0, "position": (1.5, 3.0, 2.0)},
                  {"type": MV.LIGHT_POINT, "color": (0.3, 0.5, 1.0), "intensity": 4.0, "position": (-2.0, 1.0, -1.0)},
                  {"type": MV.LIGHT_POINT, "color": (1.0, 1.0, 1.0), "intensity": 2.0, "position": (0.0, -3.0, 0.5)}]
    st = MV.MeshViewerState()
    MV.loadScene(st, gltf)
    MV.initCameraState(st)
    MV.initLaunchParams(st, w, h, lights)
    out = MV.OutputBuffer(w, h, st.device)
    tot_c = tot_s = 0
    for _ in range(subframes):
        MV.updateState(st)
        MV.launchSubframe(out, st)
        s = st.context.launch_stats()
        tot_c += s["closest_rays"]; tot_s += s["shadow_rays"]
        st.params.subframe_index += 1
    nodes, tris = st.context.accel_export(st.handle)
    sc, mats, lts, mat, normals = _oracle_for_state(st, nodes, tris)
    p = st.params
    acc = np.zeros((h, w, 4), f32)
    oc = osr = 0
    for sf in range(subframes):
        la = {"width": w, "height": h, "subframe": sf, "eye": list(p.eye), "U": list(p.U), "V": list(p.V), "W": list(p.W),
              "miss_color": list(p.miss_color)}
        acc, frame, ost = sc.render_whitted(la, lts, mats, mat, normals4=normals, accum=acc)
        oc += ost["closest_rays"]; osr += ost["shadow_rays"]
    got = st.accum.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(out.getHostPointer(), frame)
    assert tot_c == oc == w * h * subframes and tot_s == osr > 0
    hit = np.abs(acc[..., :3] - np.asarray(list(p.miss_color), f32)).sum(2) > 0
    assert 0.02 < hit.mean() and np.isfinite(acc).all() and acc[..., :3].max() > 0.05      # single-sided fixture meshes: back faces are culled
    MV.cleanup(st)


@pytest.mark.gpu
def test_whitted_launch_errors():
    import torch
    assert torch.cuda.is_available()
    from optix_raytracer_amd import _lib as L, api
    ctx = api.Context(0)
    p = L.MiptWhittedLaunchParams()
    p.width, p.height = 8, 8
    rc = ctx._lib.mipt_whitted_launch(ctx._h, None, C.byref(p), 8, 8, 1)
    assert rc == 1 and b"accum_buffer" in ctx._lib.mipt_last_error_string(ctx._h)
    acc = torch.zeros((8, 8, 4), device="cuda")
    p.accum_buffer = acc.data_ptr()
    assert ctx._lib.mipt_whitted_launch(ctx._h, None, C.byref(p), 8, 4, 1) == 1            # size mismatch
    assert ctx._lib.mipt_whitted_launch(ctx._h, None, C.byref(p), 8, 8, 1) == 6            # geometry not set
    ctx.destroy()


# ------------------------------------------------------------------------------------------------ textures
def _tex(texels, address_s=0, address_t=0, filter=1):
    return {"texels": np.ascontiguousarray(texels), "address_s": address_s, "address_t": address_t, "filter": filter}


def _sample(t, u, v):
    d = O.OrcTextureW()
    tx = t["texels"]
    d.height, d.width, d.bits, d.texels = tx.shape[0], tx.shape[1], 16 if tx.dtype == np.uint16 else 8, tx.ctypes.data
    d.address_s, d.address_t, d.filter = t["address_s"], t["address_t"], t["filter"]
    out = (C.c_float * 4)()
    O.lib().orc_sample_texture(C.byref(d), float(u), float(v), C.byref(out))
    return np.array(out, np.float32)


def test_texture_fetch_known_answers():
    tx = np.zeros((2, 4, 4), np.uint8)
    tx[0, :, 0] = (0, 51, 102, 255); tx[1, :, 0] = (255, 204, 153, 0)
    tx[..., 3] = 255
    lin = _tex(tx)
    # texel centres return the texel, between two centres their mean, past the edge the wrapped neighbour
    assert np.allclose(_sample(lin, 0.375, 0.25)[0], 51 / 255, atol=1e-7)
    assert np.allclose(_sample(lin, 0.5, 0.25)[0], (51 + 102) / 510, atol=1e-7)
    assert np.allclose(_sample(lin, 0.0, 0.25)[0], (0 + 255) / 510, atol=1e-7)          # wraps to texel 3
    assert np.allclose(_sample(lin, 0.375, 0.5)[0], (51 + 204) / 510, atol=1e-7)
    assert np.allclose(_sample(_tex(tx, address_s=1), 0.0, 0.25)[0], 0.0, atol=1e-7)       # clamp: texel 0 twice
    assert np.allclose(_sample(_tex(tx, address_s=2), -0.125, 0.25)[0], 0.0, atol=1e-7)    # mirror: index -1 -> 0
    assert np.allclose(_sample(_tex(tx, address_s=2), 1.125, 0.25)[0], 1.0, atol=1e-7)     # mirror: index 4 -> 3
    pt = _tex(tx, filter=0)
    assert _sample(pt, 0.49, 0.4)[0] == np.float32(51 / 255) and _sample(pt, 0.51, 0.6)[0] == np.float32(153 / 255)
    assert _sample(pt, 1.26, 0.1)[0] == np.float32(51 / 255)                             # wraps
    t16 = _tex((np.arange(8, dtype=np.uint16).reshape(1, 2, 4) * 8191))
    assert np.allclose(_sample(t16, 0.25, 0.5), np.arange(4) * 8191 / 65535, atol=1e-7)


def test_pow22_accuracy():
    L = O.lib()
    xs = np.concatenate([np.linspace(1 / 255, 1, 400), [1e-3, 1e-6, 0.5, 1.0]]).astype(np.float32)
    got = np.array([L.orc_det_pow22(float(x)) for x in xs])
    assert np.allclose(got, xs.astype(np.float64) ** 2.2, rtol=3e-6)
    assert L.orc_det_pow22(0.0) == 0.0 and L.orc_det_pow22(1.0) == 1.0


def test_oracle_textured_shading_known_answers():
    """a unit quad seen from above under one ambient light: the pixel is ambient * base colour, so every texture
    stage can be read off the image"""
    verts = np.zeros((6, 4), f32)
    verts[:, :3] = _quad((0, 0, 0), (0, 0, 1), (1, 0, 1), (1, 0, 0))
    uv = np.asarray(_quad((0, 0), (0, 1), (1, 1), (1, 0)), f32)                          # u = x, v = z
    sc = O.Scene({"vertices": verts, "mat_index": np.zeros(2, np.uint32), "emission": np.zeros((1, 3), f32), "diffuse": np.ones((1, 3), f32)})
    sc.build_bvh()
    w = h = 64
    U, V, W = O.camera_uvw((0.5, 2.0, 0.5), (0.5, 0, 0.5), (0, 0, 1), 2 * np.degrees(np.arctan(0.25)), 1.0)     # sees exactly the quad
    la = {"width": w, "height": h, "eye": (0.5, 2.0, 0.5), "U": U, "V": V, "W": W, "miss_color": (0, 0, 0)}
    amb = [{"type": 1, "color": (0.5, 0.25, 1.0)}]
    tx = np.zeros((2, 2, 4), np.uint8)
    tx[0, 0] = (255, 128, 0, 255); tx[0, 1] = (0, 255, 128, 255); tx[1, 0] = (128, 0, 255, 255); tx[1, 1] = (64, 64, 64, 255)
    mat = {"base_color": (0.8, 0.6, 0.4, 1.0), "metallic": 0.0, "roughness": 1.0, "emissive_factor": (0, 0, 0),
           "base_color_tex": {"texture": 0}}
    acc, _, _ = sc.render_whitted(la, amb, [mat], np.zeros(2, np.uint32), texcoords=(uv, None), textures=[_tex(tx, filter=0)])
    px = lambda u, v: acc[int(v * h), int((1 - u) * w) if U[0] < 0 else int(u * w), :3]          # noqa: E731
    # which image column shows world x: find it from the rendered quadrant colours instead of assuming the camera handedness
    lin = lambda c: (np.asarray(c, np.float64) / 255) ** 2.2                                     # noqa: E731
    want = {k: np.asarray(amb[0]["color"]) * np.asarray(mat["base_color"][:3]) * lin(tx[k][:3]) for k in ((0, 0), (0, 1), (1, 0), (1, 1))}
    quads = {(0, 0): acc[h // 4, w // 4, :3], (0, 1): acc[h // 4, 3 * w // 4, :3], (1, 0): acc[3 * h // 4, w // 4, :3], (1, 1): acc[3 * h // 4, 3 * w // 4, :3]}
    # every rendered quadrant equals exactly one texel's colour, and all four texels appear
    used = set()
    for q, val in quads.items():
        match = [k for k, c in want.items() if np.allclose(val, c, rtol=2e-5, atol=1e-7)]
        assert len(match) == 1, (q, val)
        used.add(match[0])
    assert len(used) == 4
    # vertex colours multiply in; a 90-degree texture rotation permutes the quadrants; scale 2 with wrap repeats the image
    col = np.tile(np.asarray([0.5, 1.0, 0.25, 1.0], f32), (6, 1))
    acc2, _, _ = sc.render_whitted(la, amb, [mat], np.zeros(2, np.uint32), texcoords=(uv, None), colors4=col, textures=[_tex(tx, filter=0)])
    assert np.allclose(acc2[h // 4, w // 4, :3], acc[h // 4, w // 4, :3] * (0.5, 1.0, 0.25), rtol=1e-6)
    mat_s = dict(mat, base_color_tex={"texture": 0, "scale": (2.0, 2.0)})
    acc3, _, _ = sc.render_whitted(la, amb, [mat_s], np.zeros(2, np.uint32), texcoords=(uv, None), textures=[_tex(tx, filter=0)])
    assert np.allclose(acc3[h // 8, w // 8, :3], acc[h // 4, w // 4, :3], rtol=1e-6)
    assert np.allclose(acc3[h // 8 + h // 2, w // 8 + w // 2, :3], acc[h // 4, w // 4, :3], rtol=1e-6)
    # texcoord set 1 is used when the texture info says so; without texcoords UV = barycentrics (still textured)
    mat_1 = dict(mat, base_color_tex={"texture": 0, "texcoord": 1})
    acc4, _, _ = sc.render_whitted(la, amb, [mat_1], np.zeros(2, np.uint32), texcoords=(None, uv), textures=[_tex(tx, filter=0)])
    assert np.array_equal(acc4, acc)
    # emissive texture and factor
    mat_e = dict(mat, emissive_factor=(0.5, 0.5, 0.5), emissive_tex={"texture": 0})
    acc5, _, _ = sc.render_whitted(la, amb, [mat_e], np.zeros(2, np.uint32), texcoords=(uv, None), textures=[_tex(tx, filter=0)])
    d = acc5[h // 4, w // 4, :3] - acc[h // 4, w // 4, :3]
    assert any(np.allclose(d, 0.5 * tx[k][:3] / 255.0, atol=2e-6) for k in want)
    # an undecodable texture (no texels) renders untextured
    acc6, _, _ = sc.render_whitted(la, amb, [mat], np.zeros(2, np.uint32), texcoords=(uv, None), textures=[{"texels": None}])
    assert np.allclose(acc6[h // 4, w // 4, :3], np.asarray(amb[0]["color"]) * mat["base_color"][:3], rtol=1e-6)


def test_oracle_normal_map_and_metallic_roughness_texture():
    verts = np.zeros((6, 4), f32)
    verts[:, :3] = _quad((0, 0, 0), (0, 0, 1), (1, 0, 1), (1, 0, 0))
    uv = np.asarray(_quad((0, 0), (0, 1), (1, 1), (1, 0)), f32)
    sc = O.Scene({"vertices": verts, "mat_index": np.zeros(2, np.uint32), "emission": np.zeros((1, 3), f32), "diffuse": np.ones((1, 3), f32)})
    sc.build_bvh()
    w = h = 32
    U, V, W = O.camera_uvw((0.5, 2.0, 0.5), (0.5, 0, 0.5), (0, 0, 1), 2 * np.degrees(np.arctan(0.25)), 1.0)
    la = {"width": w, "height": h, "eye": (0.5, 2.0, 0.5), "U": U, "V": V, "W": W, "miss_color": (0, 0, 0)}
    light = [{"type": 0, "color": (1, 1, 1), "intensity": 4.0, "position": (0.5, 3.0, 0.5)}]
    mat = {"base_color": (0.8, 0.8, 0.8, 1.0), "metallic": 0.5, "roughness": 0.7, "emissive_factor": (0, 0, 0)}
    flat, _, _ = sc.render_whitted(la, light, [mat], np.zeros(2, np.uint32), texcoords=(uv, None))
    up = np.zeros((1, 1, 4), np.uint8); up[0, 0] = (128, 128, 255, 255)                   # (0, 0, 1) within 1/255
    acc_up, _, _ = sc.render_whitted(la, light, [dict(mat, normal_tex={"texture": 0})], np.zeros(2, np.uint32), texcoords=(uv, None),
                                     textures=[_tex(up)])
    assert np.allclose(acc_up[..., :3], flat[..., :3], rtol=2e-2)                          # nearly the geometric normal
    tilt = np.zeros((1, 1, 4), np.uint8); tilt[0, 0] = (255, 128, 128, 255)                # (1, 0, 0): along dpdu
    acc_t, _, _ = sc.render_whitted(la, light, [dict(mat, normal_tex={"texture": 0})], np.zeros(2, np.uint32), texcoords=(uv, None),
                                    textures=[_tex(tilt)])
    assert acc_t[h // 2, w // 2, :3].sum() < 0.2 * flat[h // 2, w // 2, :3].sum()         # grazing normal: almost no light from above
    # metallic-roughness texture scales the factors: (g, b) = (255, 0) -> roughness 0.7, metallic 0
    mr = np.zeros((1, 1, 4), np.uint8); mr[0, 0] = (0, 255, 0, 255)
    acc_mr, _, _ = sc.render_whitted(la, light, [dict(mat, metallic_roughness_tex={"texture": 0})], np.zeros(2, np.uint32),
                                     texcoords=(uv, None), textures=[_tex(mr)])
    ref, _, _ = sc.render_whitted(la, light, [dict(mat, metallic=0.0)], np.zeros(2, np.uint32), texcoords=(uv, None))
    assert np.array_equal(acc_mr, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,subframes", [(150, 100, 1), (96, 64, 2)])
def test_textured_whitted_programs_match_the_oracle(tmp_path, w, h, subframes):
    import torch
    assert torch.cuda.is_available()
    from optix_raytracer_amd import meshviewer as MV
    gltf = FX.write_textured(tmp_path)
    lights = [{"type": MV.LIGHT_AMBIENT, "color": (0.05, 0.05, 0.06)},
              {"type": MV.LIGHT_POINT, "color": (1.0, 0.95, 0.9), "intensity": 9.0, "position": (1.0, 4.0, 2.5)},
              {"type": MV.LIGHT_POINT, "color": (0.4, 0.6, 1.0), "intensity": 5.0, "position": (-2.5, 1.5, 1.0)}]
    st = MV.MeshViewerState()
    MV.loadScene(st, gltf)
    assert st.d_texcoords[0] is not None and st.d_texcoords[1] is not None and st.d_colors is not None and st.textures[1] == 5
    MV.initCameraState(st)
    MV.initLaunchParams(st, w, h, lights)
    out = MV.OutputBuffer(w, h, st.device)
    for _ in range(subframes):
        MV.updateState(st)
        MV.launchSubframe(out, st)
        st.params.subframe_index += 1
    nodes, tris = st.context.accel_export(st.handle)
    sc, mats, lts, mat, normals = _oracle_for_state(st, nodes, tris)
    smats = st.scene.materials()
    for m, sm in zip(mats, smats):
        for k in ("base_color_tex", "metallic_roughness_tex", "normal_tex", "emissive_tex"):
            m[k] = sm[k]
    textures = st.scene.textures()
    assert all(t["texels"] is not None for t in textures) and textures[2]["bits"] == 16
    uv = [st.scene.flatten_texcoords(j) for j in range(2)]
    col = st.scene.flatten_colors()
    p = st.params
    acc = np.zeros((h, w, 4), f32)
    for sf in range(subframes):
        la = {"width": w, "height": h, "subframe": sf, "eye": list(p.eye), "U": list(p.U), "V": list(p.V), "W": list(p.W),
              "miss_color": list(p.miss_color)}
        acc, frame, ost = sc.render_whitted(la, lts, mats, mat, normals4=normals, accum=acc, texcoords=(uv[0][0], uv[1][0]),
                                            texcoord_present=(uv[0][1], uv[1][1]), colors4=col[0], color_present=col[1], textures=textures)
    got = st.accum.cpu().numpy()
    bad = np.argwhere(got.view(np.uint32) != acc.view(np.uint32))
    assert bad.size == 0, "first differing pixel %s: gpu %s cpu %s (%d differ)" % (bad[0], got[tuple(bad[0][:2])], acc[tuple(bad[0][:2])], len(bad))
    assert np.array_equal(out.getHostPointer(), frame)
    hit = np.abs(acc[..., :3] - np.asarray(list(p.miss_color), f32)).sum(2) > 0
    assert hit.mean() > 0.3 and np.isfinite(acc).all()
    # the textures matter: an untextured render of the same scene differs on most hit pixels
    plain, _, _ = sc.render_whitted(dict(la, subframe=0), lts, [dict(m, base_color_tex=None, normal_tex=None, emissive_tex=None,
                                                                    metallic_roughness_tex=None) for m in mats], mat, normals4=normals)
    MV.cleanup(st)
    if subframes == 1:
        assert (np.abs(plain[..., :3] - acc[..., :3]).sum(2)[hit] > 1e-3).mean() > 0.5


# ------------------------------------------------------------------------------------------------ alpha modes, face culling
def _pane_scene(extra_quads=()):
    """ground + quads at given heights, all facing +y, seen from straight above"""
    quads = [_quad((-2, 0, -2), (-2, 0, 2), (2, 0, 2), (2, 0, -2))] + [q for q in extra_quads]
    verts = np.zeros((6 * len(quads), 4), f32)
    verts[:, :3] = np.concatenate([np.asarray(q, f32) for q in quads])
    mat = np.repeat(np.arange(len(quads), dtype=np.uint32), 2)
    sc = O.Scene({"vertices": verts, "mat_index": np.zeros(len(mat), np.uint32), "emission": np.zeros((1, 3), f32), "diffuse": np.ones((1, 3), f32)})
    sc.build_bvh()
    uv = np.tile(np.asarray(_quad((0, 0), (0, 1), (1, 1), (1, 0)), f32), (len(quads), 1))
    return sc, mat, uv


def _top_launch(w, h, half=1.0, height=4.0):
    U, V, W = O.camera_uvw((0, height, 0), (0, 0, 0), (0, 0, -1), 2 * np.degrees(np.arctan(half / height)), 1.0)
    return {"width": w, "height": h, "eye": (0, height, 0), "U": U, "V": V, "W": W, "miss_color": (0.02, 0.03, 0.04)}


def test_oracle_back_face_culling_follows_the_geometry_flags():
    up = _quad((-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1))            # normal +y: front face for the camera above
    down = [up[0], up[2], up[1], up[3], up[5], up[4]]                       # reversed winding: back face
    amb = [{"type": 1, "color": (1, 1, 1)}]
    mats = [{"base_color": (0.1, 0.1, 0.1, 1), "metallic": 0, "roughness": 1, "emissive_factor": (0, 0, 0)},
            {"base_color": (0.9, 0.2, 0.2, 1), "metallic": 0, "roughness": 1, "emissive_factor": (0, 0, 0)}]
    la = _top_launch(16, 16, half=0.5)
    for quad_, flag, seen in ((up, 1, True), (down, 1, False), (down, 1 | 4, True)):
        sc, mat, uv = _pane_scene([quad_])
        sc.set_prim_flags(np.asarray([1, 1, flag, flag], np.uint32))
        acc, _, st = sc.render_whitted(la, amb, mats, mat)
        want = mats[1]["base_color"][:3] if seen else mats[0]["base_color"][:3]
        assert np.allclose(acc[8, 8, :3], want, rtol=1e-6), (flag, seen)
    # occlusion rays never cull: the back-facing pane still shadows the ground
    sc, mat, uv = _pane_scene([down])
    sc.set_prim_flags(np.asarray([1, 1, 1, 1], np.uint32))
    light = [{"type": 0, "color": (1, 1, 1), "intensity": 5.0, "position": (0, 3, 0)}]
    lit_sc, lit_mat, _ = _pane_scene([])
    lit_sc.set_prim_flags(np.asarray([1, 1], np.uint32))
    a_shadow, _, _ = sc.render_whitted(la, light, mats, mat)
    a_lit, _, _ = lit_sc.render_whitted(la, light, mats, lit_mat)
    assert a_lit[8, 8, :3].sum() > 0.01 and np.allclose(a_shadow[8, 8, :3], 0.0, atol=1e-7)


def test_oracle_alpha_mask_and_blend_known_answers():
    pane = _quad((-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1))
    la = _top_launch(32, 32, half=0.9)
    amb = [{"type": 1, "color": (1, 1, 1)}]
    ground = {"base_color": (0.2, 0.4, 0.6, 1), "metallic": 0, "roughness": 1, "emissive_factor": (0, 0, 0)}
    tx = np.zeros((1, 2, 4), np.uint8); tx[0, 0] = (255, 255, 255, 255); tx[0, 1] = (255, 255, 255, 51)       # alpha 1 | 0.2
    tex = [_tex(tx, filter=0)]
    sc, mat, uv = _pane_scene([pane])
    # ---- MASK: the half with alpha 0.2 < cutoff is a hole
    mask = {"base_color": (1, 0.5, 0.25, 1), "metallic": 0, "roughness": 1, "emissive_factor": (0, 0, 0), "alpha_mode": 1, "alpha_cutoff": 0.5,
            "base_color_tex": {"texture": 0}}
    sc.set_prim_flags(np.asarray([1, 1, 0, 0], np.uint32))
    acc, _, _ = sc.render_whitted(la, amb, [ground, mask], mat, texcoords=(uv, None), textures=tex)
    cols = {tuple(round(float(v), 4) for v in acc[16, x, :3]) for x in (4, 27)}
    assert cols == {(1.0, 0.5, 0.25), (0.2, 0.4, 0.6)}
    # ---- BLEND: local * alpha + behind * (1 - alpha); the any-hit program never rejects (cutoff 0)
    blend = dict(mask, alpha_mode=2, alpha_cutoff=0.0)
    sc.set_prim_flags(np.asarray([1, 1, 2, 2], np.uint32))
    acc, _, st = sc.render_whitted(la, amb, [ground, blend], mat, texcoords=(uv, None), textures=tex)
    got = sorted([acc[16, x, :3] for x in (4, 27)], key=lambda c: -c[0])
    a = np.float32(51 / 255)
    assert np.allclose(got[0], (1.0, 0.5, 0.25), rtol=1e-6)                                           # alpha 1: nothing shows through
    assert np.allclose(got[1], np.asarray((1.0, 0.5, 0.25)) * a + np.asarray((0.2, 0.4, 0.6)) * (1 - a), rtol=1e-6)
    assert st["closest_rays"] == 32 * 32 * 2                                                          # one continuation ray per pane hit
    # ---- shadows: a point light above; attenuation (1 - alpha) through the blend pane, none through the mask's solid half
    light = [{"type": 0, "color": (1, 1, 1), "intensity": 6.0, "position": (0.0, 3.0, 0.0)}]
    none_sc, none_mat, _ = _pane_scene([])
    none_sc.set_prim_flags(np.asarray([1, 1], np.uint32))
    free, _, _ = none_sc.render_whitted(_top_launch(32, 32, half=0.4), light, [ground], none_mat)
    sc.set_prim_flags(np.asarray([1, 1, 2, 2], np.uint32))
    # look at the ground from below the pane: move the camera under it
    U, V, W = O.camera_uvw((0, 0.9, 0), (0, 0, 0), (0, 0, -1), 2 * np.degrees(np.arctan(0.4 / 0.9)), 1.0)
    la2 = {"width": 32, "height": 32, "eye": (0, 0.9, 0), "U": U, "V": V, "W": W, "miss_color": (0, 0, 0)}
    U, V, W = O.camera_uvw((0, 0.9, 0), (0, 0, 0), (0, 0, -1), 2 * np.degrees(np.arctan(0.4 / 0.9)), 1.0)
    free2, _, _ = none_sc.render_whitted(la2, light, [ground], none_mat)
    shaded, _, _ = sc.render_whitted(la2, light, [ground, blend], mat, texcoords=(uv, None), textures=tex)
    ratio = shaded[16, :, 0] / free2[16, :, 0]
    assert np.allclose(sorted(set(np.round(ratio, 3))), [0.0, 0.8], atol=2e-3)                        # (1 - 1) and (1 - 0.2)
    sc.set_prim_flags(np.asarray([1, 1, 0, 0], np.uint32))
    shaded_m, _, _ = sc.render_whitted(la2, light, [ground, mask], mat, texcoords=(uv, None), textures=tex)
    ratio_m = shaded_m[16, :, 0] / free2[16, :, 0]
    assert np.allclose(sorted(set(np.round(ratio_m, 3))), [0.0, 1.0], atol=2e-3)                      # solid half blocks, the hole lets all light pass
    assert free[16, 16, 0] > 0


def test_oracle_blend_depth_limit():
    """nine stacked blend panes: the recursion stops at MAX_TRACE_DEPTH = 8, the ninth and the ground are never reached"""
    panes = [_quad((-1, 0.2 * k, -1), (-1, 0.2 * k, 1), (1, 0.2 * k, 1), (1, 0.2 * k, -1)) for k in range(9, 0, -1)]
    sc, mat, uv = _pane_scene(panes)
    sc.set_prim_flags(np.asarray([1, 1] + [2] * 18, np.uint32))
    amb = [{"type": 1, "color": (1, 1, 1)}]
    ground = {"base_color": (0, 1, 0, 1), "metallic": 0, "roughness": 1, "emissive_factor": (0, 0, 0)}
    film = {"base_color": (1, 0, 0, 0.5), "metallic": 0, "roughness": 1, "emissive_factor": (0, 0, 0), "alpha_mode": 2}
    acc, _, st = sc.render_whitted(_top_launch(8, 8, half=0.5), amb, [ground] + [film] * 9, mat)
    # level k contributes 0.5^k of red; after 8 levels the chain ends without the remaining transmittance
    assert np.allclose(acc[4, 4, :3], (sum(0.5 ** k for k in range(1, 9)), 0.0, 0.0), rtol=1e-6)
    assert st["closest_rays"] == 8 * 8 * 8


def _alpha_frames_match_the_oracle(gltf, w, h, subframes, persistent=None, deep=None, spatial=False):
    import torch
    assert torch.cuda.is_available()
    from optix_raytracer_amd import meshviewer as MV
    from optix_raytracer_amd import _lib as L
    lights = [{"type": MV.LIGHT_AMBIENT, "color": (0.08, 0.08, 0.1)},
              {"type": MV.LIGHT_POINT, "color": (1.0, 0.95, 0.9), "intensity": 12.0, "position": (0.5, 4.0, 4.0)},
              {"type": MV.LIGHT_POINT, "color": (0.5, 0.7, 1.0), "intensity": 6.0, "position": (-2.5, 2.5, 2.0)}]
    st = MV.MeshViewerState()
    MV.loadScene(st, gltf)
    if persistent is not None:
        st.context.set_option(L.OPT_PERSISTENT, persistent)
    MV.initCameraState(st)
    MV.initLaunchParams(st, w, h, lights)
    out = MV.OutputBuffer(w, h, st.device)
    tot_c = tot_s = 0
    for _ in range(subframes):
        MV.updateState(st)
        MV.launchSubframe(out, st)
        s = st.context.launch_stats()
        tot_c += s["closest_rays"]; tot_s += s["shadow_rays"]
        st.params.subframe_index += 1
    if deep:
        assert st.context.accel_info(st.handle)["depth"] > 8                   # the entry-pool form of the any-hit kernels ran (wh_trace_on_pool)
    if spatial:          # the tree holds more triangle records than the scene has triangles: some are referenced from several leaves
        assert st.context.accel_info(st.handle)["num_triangles"] > st.scene.info()["num_triangles"]
    nodes, tris = st.context.accel_export(st.handle)
    sc, mats, lts, mat, normals = _oracle_for_state(st, nodes, tris)
    for m, sm in zip(mats, st.scene.materials()):
        for k in ("base_color_tex", "metallic_roughness_tex", "normal_tex", "emissive_tex"):
            m[k] = sm[k]
    textures = st.scene.textures()
    uv = [st.scene.flatten_texcoords(j) for j in range(2)]
    p = st.params
    acc = np.zeros((h, w, 4), f32)
    oc = osr = 0
    for sf in range(subframes):
        la = {"width": w, "height": h, "subframe": sf, "eye": list(p.eye), "U": list(p.U), "V": list(p.V), "W": list(p.W),
              "miss_color": list(p.miss_color)}
        acc, frame, ost = sc.render_whitted(la, lts, mats, mat, normals4=normals, accum=acc, texcoords=(uv[0][0], None),
                                            texcoord_present=(uv[0][1], None), textures=textures)
        oc += ost["closest_rays"]; osr += ost["shadow_rays"]
    got = st.accum.cpu().numpy()
    bad = np.argwhere(got.view(np.uint32) != acc.view(np.uint32))
    assert bad.size == 0, "first differing pixel %s: gpu %s cpu %s (%d differ)" % (bad[0], got[tuple(bad[0][:2])], acc[tuple(bad[0][:2])], len(bad))
    assert np.array_equal(out.getHostPointer(), frame)
    assert tot_c == oc > w * h * subframes and tot_s == osr > 0                      # continuation rays were traced
    MV.cleanup(st)
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,subframes", [(150, 100, 1), (90, 60, 2)])
def test_alpha_modes_and_culling_match_the_oracle(tmp_path, w, h, subframes):
    _alpha_frames_match_the_oracle(FX.write_alpha(tmp_path), w, h, subframes)


@pytest.mark.gpu
@pytest.mark.parametrize("stack,tess,persistent,node_format", [(7, 1, None, 2), (24, 1, None, 2), (7, 24, None, 2), (24, 24, None, 2), (24, 24, 2, 2),
                                                               (24, 1, None, 0), (24, 24, None, 0), (24, 24, None, 1)])
def test_deferred_any_hit_programs_over_many_layers(tmp_path, monkeypatch, stack, tess, persistent, node_format):
    """The any-hit programs run deferred, a ray's candidates in the order of (t, primitive index) (whitted_kernels.hip): a closest-hit
    traversal finds the next candidate, the program runs on it, and an ignored candidate becomes the continuation key (tmin = t; at
    equal t only primitives above the ignored one count).  7 / 24 more panes, every fifth one TWICE -- coincident surfaces, candidates
    at bit-equal t -- so occlusion rays from the ground cross up to 30 attenuating surfaces and camera rays run into the depth limit.
    tess = 24: the ground as 1 152 triangles, a tree deep enough for the ENTRY-POOL form (k_wh_trace_pool; trees of up to 8 levels take
    the one-ray-per-thread form by themselves); persistent = 2 pins that form for the deep tree too.  Every frame equals the oracle's --
    which COLLECTS the candidates of a ray, SORTS them and multiplies -- bit for bit.
    node_format 0: fp32 boxes.  The panes are flat, so their boxes have NO thickness, and a traversal that starts at the t of an ignored
    candidate stands within rounding of the coincident twin's box: the box tests of these kernels use a slightly wider interval
    (ah_box_interval) -- without it 300-800 pixels of these frames differ.  (Formats 1 / 2 round boxes outward to fp16 and hide it.)"""
    monkeypatch.setenv("MIPT_NODE_FORMAT", str(node_format))
    _alpha_frames_match_the_oracle(FX.write_alpha(tmp_path, stack=stack, tess=tess), 120, 80, 2, persistent=persistent, deep=tess > 1)


@pytest.mark.gpu
@pytest.mark.parametrize("stack,tess,persistent", [(24, 24, None), (24, 24, 2), (7, 1, None)])
def test_deferred_any_hit_programs_over_a_tree_with_spatial_splits(tmp_path, monkeypatch, stack, tess, persistent):
    """MIPT_OPT_BUILD = 3: a triangle that straddles a split plane is referenced from several leaves, so a ray can meet the SAME candidate
    -- the same (t, primitive) key -- through two references.  The program must run on it once: the pool's candidate levels drop a key
    that is there already (pool_trace.h), the one-ray-per-thread walk continues strictly beyond (t, primitive), and the oracle -- which
    collects, sorts and multiplies -- skips a repeated candidate.  The stacked panes over the tessellated ground (large quads among
    small triangles: where spatial splits happen), every frame against the oracle bit for bit."""
    monkeypatch.setenv("MIPT_BUILD", "3")
    monkeypatch.setenv("MIPT_SBVH_FORCE", "1")          # every valid spatial split is taken: references are shared wherever a triangle straddles a plane
    _alpha_frames_match_the_oracle(FX.write_alpha(tmp_path, stack=stack, tess=tess), 120, 80, 2, persistent=persistent, deep=tess > 1, spatial=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_deferred_any_hit_programs_random_stacks(tmp_path, seed):
    """30 panes at random heights, sizes and tilts, a third of them coincident twins of the pane before, MASK / textured BLEND / untextured
    BLEND at random, over the tessellated ground (entry-pool form; seed 4 with fp32 boxes): frames equal the oracle's bit for bit."""
    import os
    if seed == 4:
        os.environ["MIPT_NODE_FORMAT"] = "0"
    try:
        _alpha_frames_match_the_oracle(FX.write_alpha(tmp_path, stack=30, tess=20, jitter=seed), 128, 96, 2, deep=True)
    finally:
        os.environ.pop("MIPT_NODE_FORMAT", None)


@pytest.mark.parametrize("name", ["textured", "alpha"])
def test_oracle_whitted_golden_fixtures(tmp_path, name):
    """the oracle's Whitted restatement on the generated glTF fixtures equals the committed golden images
    (tests/golden/make_oracle_golden.py): a change of the oracle cannot go unnoticed"""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_oracle_golden", os.path.join(here, "golden", "make_oracle_golden.py"))
    gold = np.load(os.path.join(here, "golden", "oracle_whitted_%s_48x32.npz" % name))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    acc, frame, st = mod.whitted_fixture({"textured": FX.write_textured, "alpha": FX.write_alpha}[name](tmp_path), 48, 32)
    assert np.array_equal(acc.view(np.uint32), gold["accum"].view(np.uint32))
    assert np.array_equal(frame, gold["frame"])
    assert st["closest_rays"] == int(gold["closest_rays"]) and st["shadow_rays"] == int(gold["shadow_rays"])
