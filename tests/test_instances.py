"""Two-level (instance) acceleration structures: mipt_accel_build_instances over per-mesh object-space BVHs (what
Scene::buildMeshAccels + buildInstanceAccel build, SDK/sutil/Scene.cpp:817-1212) against (a) the oracle's restatement of the
two-level walk on the SAME exported structure -- bit for bit, the parity bar -- and (b) the same scene flattened to world-space
triangles under one BVH, where only the roundings of the instance transform differ."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

# Flattened: a vertex is rounded once after object->world (m4_apply in the loader).  Two-level: the ray's origin and direction are
# rounded once after world->object, and the world->object matrix itself is an fp32 rounding of the exact inverse.  With
# coordinates of a few units (the scenes here span +-9) each of these is a relative 2^-24 .. 2^-22 perturbation of the geometry
# a ray meets; a hit distance moves by that much times the distance, and a handful of ulps more through the division in the
# triangle test.  32 ulps of t (4e-6 relative) is the bound asserted for 99.9 % of the rays; a GRAZING ray divides the same
# perturbation by the cosine of its angle to the triangle's normal, so the worst ray gets 2e-4.  What no bound on t covers is a
# ray within that distance of a triangle EDGE, which then hits the neighbouring triangle or misses -- counted, and required to
# stay below 3 in 10 000 rays.
T_REL_TOL = 4e-6
T_REL_TOL_GRAZING = 2e-4
EDGE_FLIPS = 3e-4


def _build(ctx, torch, scene_file):
    """-> (flattened scene dict, handle of the two-level structure, keep-alive list)"""
    from optix_raytracer_amd import api, scenes
    gl = api.Scene(scene_file)
    flat = scenes.from_gltf(scene_file)
    meshes, insts = scenes.gltf_two_level(gl)
    keep, gas = [], []
    n_mat = flat["diffuse"].shape[0]
    for verts, mat in meshes:
        dv = torch.from_numpy(np.ascontiguousarray(verts)).cuda()
        dm = torch.from_numpy(np.where(mat == 0xFFFFFFFF, n_mat - 1, mat).astype(np.int32)).cuda()
        keep += [dv, dm]
        gas.append(ctx.accel_build([{"vertices": dv.data_ptr(), "num_vertices": verts.shape[0], "stride": 16, "sbt_index_offsets": dm.data_ptr()}]))
    inst = [(gas[m], xf) for m, xf in insts]
    ias = ctx.accel_build_instances(inst)
    return flat, ias, gas, inst, keep


def _oracle_two_level(ctx, flat, ias):
    nodes, order, inv, base, gas_handles = ctx.accel_export_instances(ias)
    exports = {}
    for h in set(int(x) for x in gas_handles):
        exports[h] = ctx.accel_export(h)
    osc = O.Scene(flat)
    osc.set_instances(nodes, order, inv, base, [exports[int(h)] for h in gas_handles])
    return osc


def _state(ctx, torch, flat, handle, w, h, spp, depth):
    """launch state over `handle`; the hit-group records point at the FLATTENED world-space vertices (a hit reports the flattened index)"""
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT
    st = PT.PathTracerState()
    st.device = torch.device("cuda", 0)
    st.context = ctx
    st.scene = flat
    st.d_vertices = torch.from_numpy(np.ascontiguousarray(flat["vertices"], np.float32)).cuda()
    st.gas_handle = handle
    PT.createSBT(st)
    PT.initLaunchParams(st, w, h, spp)
    cam = flat["camera"]
    PT.handleCameraUpdate(st, PT.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], 1.0))
    ctx.set_option(L.OPT_MAX_DEPTH, depth)
    return PT, st


def _render(PT, st):
    ob = PT.OutputBuffer(st.params.width, st.params.height, st.device)
    PT.launchSubframe(ob, st)
    return st.accum.cpu().numpy(), ob.getHostPointer(), st.context.launch_stats()


@pytest.mark.parametrize("builder", [None, 4])
def test_two_level_structure_matches_the_oracle_and_the_flattened_scene(tmp_path, monkeypatch, builder):
    """builder = 4: the meshes built on the device with spatial splits (MIPT_OPT_BUILD = 4) -- their leaves hold more triangle records
    than the meshes have triangles, the hit records are the same"""
    import torch
    assert torch.cuda.is_available()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import api, scenes
    path = scenes.write_instanced_gltf(tmp_path / "instanced.gltf")
    if builder is not None:
        monkeypatch.setenv("MIPT_BUILD", str(builder)); monkeypatch.setenv("MIPT_BUILD_HOST_BELOW", "0")
    ctx = api.Context(0)
    flat, ias, gas, inst, keep = _build(ctx, torch, path)
    info = ctx.accel_info(ias)
    assert info["num_instances"] == len(inst) == 151 and info["num_triangles"] == flat["vertices"].shape[0] // 3
    if builder == 4:
        meshes = scenes.gltf_two_level(api.Scene(path))[0]
        assert sum(ctx.accel_info(g)["num_triangles"] for g in gas) > sum(v.shape[0] // 3 for v, _ in meshes)
    w, h, spp, depth = 320, 180, 2, 4
    launch = scenes.launch_desc(flat, w, h, spp)
    # ---- (a) GPU over the two-level structure == the oracle's two-level walk, bit for bit (frames, counters, visit counts)
    PT, st = _state(ctx, torch, flat, ias, w, h, spp, depth)
    acc, frame, stats = _render(PT, st)
    osc = _oracle_two_level(ctx, flat, ias)
    a, f, ost = osc.render(launch, mode=O.MODE_INST, max_depth=depth, threads=8)
    assert np.array_equal(acc.view(np.uint32), a.view(np.uint32)) and np.array_equal(frame, f)
    assert stats["closest_rays"] == ost["closest_rays"] and stats["shadow_rays"] == ost["shadow_rays"] > 1000
    ctx.set_option(L.OPT_COUNT_VISITS, 1)
    acc_c, _, stats_c = _render(PT, st)
    ctx.set_option(L.OPT_COUNT_VISITS, 0)
    assert np.array_equal(acc_c.view(np.uint32), a.view(np.uint32))
    assert stats_c["node_visits"] == ost["node_visits"] and stats_c["tri_tests"] == ost["tri_tests"]
    # ---- ray-buffer queries: hit records and occlusion flags equal the oracle's
    rays = osc.primary_rays(scenes.launch_desc(flat, 640, 360))
    rng = np.random.default_rng(5)
    extra = np.zeros((20000, 8), np.float32)
    extra[:, :3] = rng.uniform(-8, 8, (20000, 3)) * (1, 0.4, 1) + (0, 1.5, 0)
    extra[:, 4:7] = rng.normal(size=(20000, 3))
    extra[:, 3], extra[:, 7] = 0.01, 1e16
    extra[:100, 7] = 0.0
    rays = np.concatenate([rays, extra])
    n = rays.shape[0]
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    ctx.trace_closest(ias, d_rays.data_ptr(), n, d_hits.data_ptr())
    ctx.sync()
    hits2 = d_hits.cpu().numpy()
    t, prim, u, v = osc.trace_closest(rays, O.MODE_INST)
    assert np.array_equal(hits2[:, 1].view(np.uint32), prim) and np.array_equal(hits2[:, 0].view(np.uint32), t.view(np.uint32))
    hit = prim != 0xffffffff
    assert np.array_equal(hits2[hit, 2].view(np.uint32), u[hit].view(np.uint32)) and np.array_equal(hits2[hit, 3].view(np.uint32), v[hit].view(np.uint32))
    seg = rays.copy()
    seg[:, 7] = np.where(seg[:, 7] > 1.0, 6.0, seg[:, 7])
    d_seg = torch.from_numpy(seg).cuda()
    d_occ = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.trace_any(ias, d_seg.data_ptr(), n, d_occ.data_ptr())
    assert np.array_equal(d_occ.cpu().numpy().astype(np.uint8), osc.trace_any(seg, O.MODE_INST))
    # ---- (b) against the flattened scene under ONE BVH: the same triangles up to the rounding of the instance transform
    dv = torch.from_numpy(np.ascontiguousarray(flat["vertices"], np.float32)).cuda()
    dm = torch.from_numpy(flat["mat_index"].astype(np.int32)).cuda()
    one = ctx.accel_build([{"vertices": dv.data_ptr(), "num_vertices": flat["vertices"].shape[0], "stride": 16, "sbt_index_offsets": dm.data_ptr()}])
    ctx.trace_closest(one, d_rays.data_ptr(), n, d_hits.data_ptr())
    ctx.sync()
    hits1 = d_hits.cpu().numpy()
    p1, p2 = hits1[:, 1].view(np.uint32), hits2[:, 1].view(np.uint32)
    same = p1 == p2
    both = same & (p1 != 0xffffffff)
    assert (~same).mean() <= EDGE_FLIPS, "%d of %d rays meet another triangle" % ((~same).sum(), n)
    rel = np.abs(hits1[both, 0] - hits2[both, 0]) / np.maximum(np.abs(hits1[both, 0]), 1.0)
    assert np.quantile(rel, 0.999) <= T_REL_TOL and rel.max() <= T_REL_TOL_GRAZING, (np.quantile(rel, 0.999), rel.max())
    ctx.trace_any(one, d_seg.data_ptr(), n, d_occ.data_ptr())
    assert (d_occ.cpu().numpy().astype(np.uint8) != osc.trace_any(seg, O.MODE_INST)).mean() <= EDGE_FLIPS
    PT1, st1 = _state(ctx, torch, flat, one, w, h, spp, depth)
    acc1, frame1, stats1 = _render(PT1, st1)
    # frames: a path that meets the same triangles differs by rounding only; a path that flips at an edge differs freely
    close = np.isclose(acc1[..., :3], acc[..., :3], rtol=2e-3, atol=2e-4).all(-1)
    assert close.mean() >= 0.995, close.mean()
    assert abs(int(stats1["shadow_rays"]) - int(stats["shadow_rays"])) <= 0.002 * stats["shadow_rays"]
    print("two-level vs flattened: %d / %d rays meet another triangle, max relative dt %.2e, %.4f of the pixels agree to 2e-3" % (
        (~same).sum(), n, rel.max(), close.mean()))
    # ---- an instance moves: mipt_accel_update_instances == a fresh build with the new transforms
    moved = list(inst)
    xf = np.array(moved[7][1], np.float32, copy=True)
    xf[:, 3] += np.float32([0.8, 0.6, -0.5])
    moved[7] = (moved[7][0], xf)
    ctx.accel_update_instances(ias, moved)
    fresh = ctx.accel_build_instances(moved)
    ctx.trace_closest(ias, d_rays.data_ptr(), n, d_hits.data_ptr()); ctx.sync()
    h_upd = d_hits.cpu().numpy()
    ctx.trace_closest(fresh, d_rays.data_ptr(), n, d_hits.data_ptr()); ctx.sync()
    assert np.array_equal(h_upd.view(np.uint32), d_hits.cpu().numpy().view(np.uint32))
    assert not np.array_equal(h_upd.view(np.uint32), hits2.view(np.uint32))
    t_m, prim_m, _, _ = _oracle_two_level(ctx, flat, ias).trace_closest(rays, O.MODE_INST)
    assert np.array_equal(h_upd[:, 1].view(np.uint32), prim_m) and np.array_equal(h_upd[:, 0].view(np.uint32), t_m.view(np.uint32))
    # ---- errors: a mesh BVH cannot go away while an instance structure uses it; a singular transform is refused
    with pytest.raises(api.MiptError):
        ctx.accel_destroy(gas[0])
    bad = list(inst)
    bad[3] = (bad[3][0], np.zeros((3, 4), np.float32))
    with pytest.raises(api.MiptError):
        ctx.accel_build_instances(bad)
    with pytest.raises(api.MiptError):
        ctx.accel_build_instances([(ias, np.eye(4, dtype=np.float32)[:3])])      # an instance structure as a child
    # ---- a FAILED update changes nothing: a singular matrix at instance k is refused, the structure still gives the
    # same hits, its export tables still describe the device copy, a later valid update works, and every structure can be destroyed
    before = ctx.accel_export_instances(ias)
    badu = list(moved)
    badu[5] = (badu[5][0], np.zeros((3, 4), np.float32))
    with pytest.raises(api.MiptError):
        ctx.accel_update_instances(ias, badu)
    after = ctx.accel_export_instances(ias)
    for b_, a_ in zip(before, after):
        assert np.array_equal(b_.view(np.uint8), a_.view(np.uint8))
    ctx.trace_closest(ias, d_rays.data_ptr(), n, d_hits.data_ptr()); ctx.sync()
    assert np.array_equal(h_upd.view(np.uint32), d_hits.cpu().numpy().view(np.uint32))
    ctx.accel_update_instances(ias, inst)                                         # back to the first placement
    ctx.trace_closest(ias, d_rays.data_ptr(), n, d_hits.data_ptr()); ctx.sync()
    assert np.array_equal(hits2.view(np.uint32), d_hits.cpu().numpy().view(np.uint32))
    ctx.accel_destroy(fresh); ctx.accel_destroy(ias)
    for g in sorted(set(gas)):
        ctx.accel_destroy(g)                                                      # no reference is left behind
    ctx.destroy()


def test_whitted_programs_over_the_instance_structure(tmp_path):
    """optixMeshViewer's pipeline as the reference builds it: one triangle structure per mesh + the instance structure
    (meshviewer.loadScene(two_level=True)); radiance rays cull back faces by the per-triangle geometry flags inside the instance
    walk, occlusion rays use the any-hit form.  Frames, frame buffer and ray counters equal the oracle's two-level render."""
    import torch
    from optix_raytracer_amd import meshviewer as MV
    from optix_raytracer_amd import scenes
    path = scenes.write_instanced_gltf(tmp_path / "instanced.gltf", n_instances=60, seed=4)
    w, h = 256, 144
    st = MV.MeshViewerState()
    MV.loadScene(st, path, two_level=True)
    MV.initCameraState(st)
    MV.initLaunchParams(st, w, h)
    out = MV.OutputBuffer(w, h, st.device)
    tot = [0, 0]
    for _ in range(2):
        MV.updateState(st)
        MV.launchSubframe(out, st)
        s = st.context.launch_stats()
        tot[0] += s["closest_rays"]; tot[1] += s["shadow_rays"]
        st.params.subframe_index += 1
    ctx = st.context
    assert st.instanced_shading and st.d_vertices is None           # the shading tables are per mesh + instance records: nothing flattened on the device
    verts, mat = st.scene.flatten()                                 # (the oracle shades from the flattened world-space arrays)
    flat = {"vertices": verts, "mat_index": np.zeros(mat.shape[0], np.uint32), "emission": np.zeros((1, 3), np.float32), "diffuse": np.ones((1, 3), np.float32)}
    osc = _oracle_two_level(ctx, flat, st.handle)
    gf = MV.geometry_flags(st.materials)
    osc.set_prim_flags(gf[np.minimum(mat, len(gf) - 1)])
    mats = [{"base_color": list(m.base_color), "metallic": m.metallic, "roughness": m.roughness, "emissive_factor": list(m.emissive_factor),
             "alpha_mode": m.alpha_mode, "alpha_cutoff": m.alpha_cutoff, "double_sided": m.double_sided} for m in st.materials]
    lights = [{"type": l.type, "color": list(l.color), "intensity": l.intensity, "position": list(l.position), "falloff": l.falloff}
              for l in list(st.lights)[: st.params.lights.count]]
    p = st.params
    acc = np.zeros((h, w, 4), np.float32)
    oc = [0, 0]
    for sf in range(2):
        la = {"width": w, "height": h, "subframe": sf, "eye": list(p.eye), "U": list(p.U), "V": list(p.V), "W": list(p.W), "miss_color": list(p.miss_color)}
        acc, frame, ost = osc.render_whitted(la, lights, mats, mat, accum=acc, mode=O.MODE_INST)
        oc[0] += ost["closest_rays"]; oc[1] += ost["shadow_rays"]
    got = st.accum.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), acc.view(np.uint32)) and np.array_equal(out.getHostPointer(), frame)
    assert tot == oc and tot[1] > 1000
    hit = np.abs(acc[..., :3] - np.asarray(list(p.miss_color), np.float32)).sum(2) > 0
    assert hit.mean() > 0.3
    MV.cleanup(st)


def test_any_hit_programs_inside_instances(tmp_path):
    """MASK and BLEND materials on the INSTANCED meshes: the Whitted any-hit programs (alpha test for radiance rays, attenuation
    for occlusion rays, cuda/whitted.cu:100-136) run over the two-level structure (k_wh_trace_pool_inst).  Checked against the same scene
    flattened under one BVH (k_wh_trace_pool, which the oracle pins in tests/test_whitted.py): only the roundings of the instance
    transform differ, so almost every pixel is bit-identical and the rest are pixels whose ray passes within rounding of a
    triangle or texel edge."""
    import torch
    import _png_fixture as PF
    from optix_raytracer_amd import meshviewer as MV
    from optix_raytracer_amd import scenes
    rng = np.random.default_rng(11)
    film = np.zeros((8, 8, 4), np.int64); film[..., :3] = rng.integers(60, 255, (8, 8, 3)); film[..., 3] = rng.integers(40, 230, (8, 8))
    fence = np.zeros((8, 8, 4), np.int64); fence[..., :3] = 200; fence[..., 3] = np.where((np.add.outer(np.arange(8), np.arange(8)) % 3) == 0, 0, 255)
    path = scenes.write_instanced_gltf(tmp_path / "instanced_alpha.gltf", n_instances=80, seed=6,
                                       alpha_textures=(PF.encode(film, 6, 8)[0], PF.encode(fence, 6, 8)[0]))
    w, h = 320, 180
    frames, stats = {}, {}
    for two_level in (True, False):
        st, img = MV.render(path, w, h, subframes=2, two_level=two_level)
        assert any(m.alpha_mode != 0 for m in st.materials)
        info = st.context.accel_info(st.handle)
        assert (info["num_instances"] > 0) == two_level
        frames[two_level] = st.accum.cpu().numpy().copy()
        stats[two_level] = st.context.launch_stats()
        MV.cleanup(st)
    a, b = frames[True], frames[False]
    assert np.isfinite(a).all() and a[..., :3].max() > 0.05
    # (an occlusion ray multiplies the (1 - alpha) of the surfaces it crosses in the order of t over both structures; what differs are
    #  the roundings of t, u, v themselves -- the object-space ray -- so products agree to rounding, not bit for bit)
    d = np.abs(a - b)[..., :3].max(-1)
    print("pixels that differ by more than 1e-5: %.4f, more than 1e-2: %.4f, mean |d| %.2e" % ((d > 1e-5).mean(), (d > 1e-2).mean(), d.mean()))
    assert (d > 1e-5).mean() < 0.02 and (d > 1e-2).mean() < 0.005 and d.mean() < 1e-3
    # the any-hit programs did run: ignored candidates make radiance rays continue, attenuated shadows are neither 0 nor 1
    assert stats[True]["closest_rays"] > w * h and abs(stats[True]["shadow_rays"] - stats[False]["shadow_rays"]) < 0.01 * stats[False]["shadow_rays"]


@pytest.mark.gpu
@pytest.mark.parametrize("persistent,seed,n_instances,builder", [(None, 8, 70, None), (2, 8, 70, None), (None, 31, 140, None), (None, 8, 70, 4), (2, 31, 140, 4)])
def test_any_hit_programs_inside_instances_match_the_oracle(tmp_path, monkeypatch, persistent, seed, n_instances, builder):
    """The deferred any-hit programs over the two-level structure (k_wh_trace_pool_inst; persistent = 2: k_wh_trace_inst, one ray per
    thread) against the ORACLE's two-level render of the same structure: textured BLEND and MASK materials on the instanced meshes,
    frames, frame buffer and ray counters bit for bit -- candidates in the order of (t, flattened primitive index), t taken with the
    object-space ray, exactly as the oracle's gas_traverse has it."""
    import torch
    import _png_fixture as PF
    from optix_raytracer_amd import meshviewer as MV
    from optix_raytracer_amd import scenes
    from optix_raytracer_amd import _lib as L
    rng = np.random.default_rng(12)
    film = np.zeros((8, 8, 4), np.int64); film[..., :3] = rng.integers(60, 255, (8, 8, 3)); film[..., 3] = rng.integers(40, 230, (8, 8))
    fence = np.zeros((8, 8, 4), np.int64); fence[..., :3] = 200; fence[..., 3] = np.where((np.add.outer(np.arange(8), np.arange(8)) % 3) == 0, 0, 255)
    path = scenes.write_instanced_gltf(tmp_path / "instanced_alpha.gltf", n_instances=n_instances, seed=seed,
                                       alpha_textures=(PF.encode(film, 6, 8)[0], PF.encode(fence, 6, 8)[0]))
    w, h = 200, 112
    if builder is not None:       # the meshes under the instances built with spatial splits: a candidate met through two references counts once
        monkeypatch.setenv("MIPT_BUILD", str(builder)); monkeypatch.setenv("MIPT_BUILD_HOST_BELOW", "0")
    st = MV.MeshViewerState()
    MV.loadScene(st, path, two_level=True)
    if persistent is not None:
        st.context.set_option(L.OPT_PERSISTENT, persistent)
    MV.initCameraState(st)
    MV.initLaunchParams(st, w, h)
    out = MV.OutputBuffer(w, h, st.device)
    tot = [0, 0]
    for _ in range(2):
        MV.updateState(st)
        MV.launchSubframe(out, st)
        s = st.context.launch_stats()
        tot[0] += s["closest_rays"]; tot[1] += s["shadow_rays"]
        st.params.subframe_index += 1
    ctx = st.context
    assert any(m.alpha_mode != 0 for m in st.materials) and ctx.accel_info(st.handle)["num_instances"] > 0
    verts, mat = st.scene.flatten()
    flat = {"vertices": verts, "mat_index": np.zeros(mat.shape[0], np.uint32), "emission": np.zeros((1, 3), np.float32), "diffuse": np.ones((1, 3), np.float32)}
    osc = _oracle_two_level(ctx, flat, st.handle)
    if builder == 4:
        gas_handles = set(int(x) for x in ctx.accel_export_instances(st.handle)[4])
        assert sum(ctx.accel_info(g)["num_triangles"] for g in gas_handles) > sum(int(ctx.accel_export(g)[1].view(np.uint32)[:, 3].max()) + 1 for g in gas_handles)
    gf = MV.geometry_flags(st.materials)
    osc.set_prim_flags(gf[np.minimum(mat, len(gf) - 1)])
    mats = [{"base_color": list(m.base_color), "metallic": m.metallic, "roughness": m.roughness, "emissive_factor": list(m.emissive_factor),
             "alpha_mode": m.alpha_mode, "alpha_cutoff": m.alpha_cutoff, "double_sided": m.double_sided} for m in st.materials]
    for m, sm in zip(mats, st.scene.materials()):
        for k in ("base_color_tex", "metallic_roughness_tex", "normal_tex", "emissive_tex"):
            m[k] = sm[k]
    lights = [{"type": l.type, "color": list(l.color), "intensity": l.intensity, "position": list(l.position), "falloff": l.falloff}
              for l in list(st.lights)[: st.params.lights.count]]
    uv = st.scene.flatten_texcoords(0)
    normals = st.scene.flatten_normals()
    p = st.params
    acc = np.zeros((h, w, 4), np.float32)
    oc = [0, 0]
    for sf in range(2):
        la = {"width": w, "height": h, "subframe": sf, "eye": list(p.eye), "U": list(p.U), "V": list(p.V), "W": list(p.W), "miss_color": list(p.miss_color)}
        acc, frame, ost = osc.render_whitted(la, lights, mats, mat, normals4=normals, accum=acc, texcoords=(uv[0], None), texcoord_present=(uv[1], None),
                                            textures=st.scene.textures(), mode=O.MODE_INST)
        oc[0] += ost["closest_rays"]; oc[1] += ost["shadow_rays"]
    got = st.accum.cpu().numpy()
    bad = np.argwhere(got.view(np.uint32) != acc.view(np.uint32))
    assert bad.size == 0, "first differing pixel %s: gpu %s cpu %s (%d differ)" % (bad[0], got[tuple(bad[0][:2])], acc[tuple(bad[0][:2])], len(bad))
    assert np.array_equal(out.getHostPointer(), frame)
    assert tot == oc and tot[0] > w * h * 2 and tot[1] > 1000                      # continuation rays were traced
    MV.cleanup(st)


@pytest.mark.gpu
@pytest.mark.parametrize("two_level", [False, True])
def test_any_hit_forms_agree_on_a_large_scene(tmp_path, two_level):
    """A scene of 300 instances / 117 k alpha-textured triangles (BLEND films + MASK fences) at 960x540, three subframes: the entry-pool
    form of the any-hit kernels (64 rays per wave, three candidates per traversal after the first, radiance rays seeded by the tiled
    launch over the two-level structure) and the one-ray-per-thread form (one candidate per traversal) must give the same frame bit for
    bit -- every candidate of every ray in the order of (t, primitive), whatever the traversal's shape."""
    import torch
    import _png_fixture as PF
    from optix_raytracer_amd import meshviewer as MV
    from optix_raytracer_amd import scenes
    from optix_raytracer_amd import _lib as L
    rng = np.random.default_rng(21)
    film = np.zeros((8, 8, 4), np.int64); film[..., :3] = rng.integers(60, 255, (8, 8, 3)); film[..., 3] = rng.integers(40, 230, (8, 8))
    fence = np.zeros((8, 8, 4), np.int64); fence[..., :3] = 200; fence[..., 3] = np.where((np.add.outer(np.arange(8), np.arange(8)) % 3) == 0, 0, 255)
    path = scenes.write_instanced_gltf(tmp_path / "instanced_alpha_large.gltf", n_instances=300, seed=5,
                                       alpha_textures=(PF.encode(film, 6, 8)[0], PF.encode(fence, 6, 8)[0]))
    w, h = 960, 540
    frames = {}
    for form in ("pool", "walk"):
        st = MV.MeshViewerState()
        MV.loadScene(st, path, two_level=two_level)
        if form == "walk":
            st.context.set_option(L.OPT_PERSISTENT, 2)
        MV.initCameraState(st)
        MV.initLaunchParams(st, w, h)
        out = MV.OutputBuffer(w, h, st.device)
        for _ in range(3):
            MV.updateState(st)
            MV.launchSubframe(out, st)
            st.params.subframe_index += 1
        assert st.context.accel_info(st.handle)["depth"] > 8
        frames[form] = (st.accum.cpu().numpy().copy(), st.context.launch_stats())
        MV.cleanup(st)
    a, b = frames["pool"][0], frames["walk"][0]
    assert np.isfinite(a).all() and a[..., :3].max() > 0.05
    bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
    assert bad.size == 0, "first differing pixel %s: pool %s walk %s (%d differ)" % (bad[0], a[tuple(bad[0][:2])], b[tuple(bad[0][:2])], len(bad))
    assert frames["pool"][1]["closest_rays"] == frames["walk"][1]["closest_rays"] > w * h
    assert frames["pool"][1]["shadow_rays"] == frames["walk"][1]["shadow_rays"] > 0


@pytest.mark.parametrize("form", [1, 2])
def test_every_traversal_form_gives_the_same_records(tmp_path, monkeypatch, form):
    """Round 3: three forms walk a two-level structure -- tiles of 64 rays, one ray per lane with the object-space ray in registers, then
    the entry pool (k_trace_tiled_inst: big closest-hit launches); the entry pool from the root (k_trace_static_inst: any-hit launches,
    small launches); one ray per lane to the end (k_trace_inst: counting launches, meshes without a BVH4 mirror).  MIPT_INST_FORM pins
    form 1 or 2 for a context; hit records and occlusion flags of 0.3 M rays must equal those of the default context (which takes the
    tiled form for the closest-hit launch) bit for bit, and the oracle's."""
    import torch
    from optix_raytracer_amd import api, scenes
    path = scenes.write_instanced_gltf(tmp_path / "instanced.gltf", n_instances=90, seed=11)
    results = []
    for env in (None, str(form)):
        if env is None:
            monkeypatch.delenv("MIPT_INST_FORM", raising=False)
        else:
            monkeypatch.setenv("MIPT_INST_FORM", env)
        ctx = api.Context(0)
        flat, ias, gas, inst, keep = _build(ctx, torch, path)
        osc = _oracle_two_level(ctx, flat, ias)
        rays = osc.primary_rays(scenes.launch_desc(flat, 720, 405))
        rng = np.random.default_rng(8)
        extra = np.zeros((30000, 8), np.float32)
        extra[:, :3] = rng.uniform(-8, 8, (30000, 3)) * (1, 0.4, 1) + (0, 1.5, 0)
        extra[:, 4:7] = rng.normal(size=(30000, 3))
        extra[:, 3], extra[:, 7] = 0.01, 1e16
        rays = np.concatenate([rays, extra])
        n = rays.shape[0]
        d_rays = torch.from_numpy(rays).cuda()
        d_hits = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
        ctx.trace_closest(ias, d_rays.data_ptr(), n, d_hits.data_ptr()); ctx.sync()
        seg = rays.copy(); seg[:, 7] = np.where(seg[:, 7] > 1.0, 5.0, seg[:, 7])
        d_seg = torch.from_numpy(seg).cuda()
        d_occ = torch.zeros(n, dtype=torch.int32, device="cuda")
        ctx.trace_any(ias, d_seg.data_ptr(), n, d_occ.data_ptr()); ctx.sync()
        results.append((d_hits.cpu().numpy(), d_occ.cpu().numpy()))
        if env is None:
            t, prim, u, v = osc.trace_closest(rays, O.MODE_INST)
            h = results[0][0]
            assert np.array_equal(h[:, 1].view(np.uint32), prim) and np.array_equal(h[:, 0].view(np.uint32), t.view(np.uint32))
            assert np.array_equal(results[0][1].astype(np.uint8), osc.trace_any(seg, O.MODE_INST))
        ctx.destroy()
    assert np.array_equal(results[0][0].view(np.uint32), results[1][0].view(np.uint32))
    assert np.array_equal(results[0][1], results[1][1])


def test_a_refitted_mesh_makes_its_instance_structures_stale(tmp_path):
    """An instance structure holds copies of its meshes' mirrors and world boxes made from their root boxes.  After mipt_accel_update of a
    mesh it must be brought up to date (OptiX asks for an IAS update after a GAS update too); until then tracing it is an error, not
    a picture from old boxes.  After mipt_accel_update_instances it equals a fresh build over the moved mesh."""
    import torch
    from optix_raytracer_amd import api, scenes
    ctx = api.Context(0)
    path = scenes.write_instanced_gltf(tmp_path / "instanced.gltf", n_instances=40, seed=5)
    flat, ias, gas, inst, keep = _build(ctx, torch, path)
    osc = _oracle_two_level(ctx, flat, ias)
    rays = osc.primary_rays(scenes.launch_desc(flat, 320, 180))
    n = rays.shape[0]
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    ctx.trace_closest(ias, d_rays.data_ptr(), n, d_hits.data_ptr()); ctx.sync()
    before = d_hits.cpu().numpy()
    # move mesh 0's vertices (same topology) and refit it
    dv, dm = keep[0], keep[1]
    v = dv.cpu().numpy().copy()
    v[:, :3] *= np.float32(1.15)
    dv.copy_(torch.from_numpy(v))
    ctx.accel_update(gas[0], [{"vertices": dv.data_ptr(), "num_vertices": v.shape[0], "stride": 16, "sbt_index_offsets": dm.data_ptr()}])
    with pytest.raises(api.MiptError) as e:
        ctx.trace_closest(ias, d_rays.data_ptr(), n, d_hits.data_ptr())
    assert "mipt_accel_update_instances" in str(e.value)
    ctx.accel_update_instances(ias, inst)
    ctx.trace_closest(ias, d_rays.data_ptr(), n, d_hits.data_ptr()); ctx.sync()
    after = d_hits.cpu().numpy()
    fresh = ctx.accel_build_instances(inst)
    ctx.trace_closest(fresh, d_rays.data_ptr(), n, d_hits.data_ptr()); ctx.sync()
    assert np.array_equal(after.view(np.uint32), d_hits.cpu().numpy().view(np.uint32))
    assert not np.array_equal(after.view(np.uint32), before.view(np.uint32))
    t, prim, _, _ = _oracle_two_level(ctx, flat, ias).trace_closest(rays, O.MODE_INST)
    assert np.array_equal(after[:, 1].view(np.uint32), prim) and np.array_equal(after[:, 0].view(np.uint32), t.view(np.uint32))
    ctx.destroy()


def test_instanced_shading_tables_give_the_flattened_frame(tmp_path):
    """Round 3: the shading tables the reference keeps (Scene::createSBT, Scene.cpp:1368-1480) -- per-mesh OBJECT-space buffers + one
    record per instance (mipt_whitted_set_geometry_instanced) -- against the flattened world-space arrays over the same instance
    structure: a corner is taken to world space in the kernel with the arithmetic mipt_scene_flatten uses on the host, a normal with
    mipt_scene_flatten_normals', so the frames are bit-identical (textured BLEND / MASK materials, normals, general affine transforms)."""
    import torch
    from optix_raytracer_amd import meshviewer as MV
    from optix_raytracer_amd import scenes
    import _png_fixture as PF
    rng = np.random.default_rng(4)
    film = np.zeros((8, 8, 4), np.int64); film[..., :3] = rng.integers(60, 255, (8, 8, 3)); film[..., 3] = rng.integers(40, 230, (8, 8))
    fence = np.zeros((8, 8, 4), np.int64); fence[..., :3] = 200; fence[..., 3] = np.where((np.add.outer(np.arange(8), np.arange(8)) % 3) == 0, 0, 255)
    png = (PF.encode(film, 6, 8)[0], PF.encode(fence, 6, 8)[0])
    frames = []
    for inst_sh in (True, False):
        path = scenes.write_instanced_gltf(tmp_path / ("instanced%d.gltf" % inst_sh), n_instances=70, seed=6, alpha_textures=png)
        st, image = MV.render(path, 320, 180, subframes=2, two_level=True, instanced_shading=inst_sh)
        assert st.instanced_shading == inst_sh
        frames.append((st.accum.cpu().numpy().copy(), image.copy(), st.context.launch_stats()))
        MV.cleanup(st)
    (a0, f0, s0), (a1, f1, s1) = frames
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32)) and np.array_equal(f0, f1)
    assert s0["closest_rays"] == s1["closest_rays"] and s0["shadow_rays"] == s1["shadow_rays"] > 0
    assert (a0[..., :3] > 0).mean() > 0.2


def test_instanced_tables_must_match_the_traversable(tmp_path):
    """mipt_whitted_set_geometry_instanced keeps one record per instance and one word per flattened primitive; a launch over a
    traversable that counts other triangles, or lists its instances in another order, would index those tables out of bounds on the
    device.  The library owns every count it needs, so mipt_whitted_launch refuses (MIPT_ERROR_INVALID_VALUE) instead."""
    from optix_raytracer_amd import api
    from optix_raytracer_amd import meshviewer as MV
    from optix_raytracer_amd import scenes
    path = scenes.write_instanced_gltf(tmp_path / "instanced.gltf", n_instances=40, seed=3)
    st, image = MV.render(path, 96, 64, two_level=True)
    assert st.instanced_shading
    out = MV.OutputBuffer(96, 64, st.device)
    good = st.params.handle
    fewer = st.context.accel_build_instances(st.instances[:-1])                   # another triangle total
    st.params.handle = fewer
    with pytest.raises(api.MiptError) as e:
        MV.launchSubframe(out, st)
    assert e.value.code == 1 and "triangles" in str(e.value)
    # the same instances in another order: the total agrees, the first flattened primitive of an instance does not
    k = next(i for i in range(1, len(st.instances)) if st.instances[i][0] != st.instances[0][0])
    swapped = list(st.instances); swapped[0], swapped[k] = swapped[k], swapped[0]
    st.params.handle = st.context.accel_build_instances(swapped)
    with pytest.raises(api.MiptError) as e:
        MV.launchSubframe(out, st)
    assert e.value.code == 1 and "instance order" in str(e.value)
    st.params.handle = good                                                        # the structure the tables were made for still renders
    st.params.subframe_index = 0
    MV.launchSubframe(out, st)
    assert np.array_equal(out.getHostPointer(), image)
    MV.cleanup(st)


def _spike_mesh(n, rng):
    """n triangles whose sizes and positions shrink geometrically: a SAH builder peels them off one or two at a time, so the tree --
    and a lane's worst-case stack over its BVH4 mirror -- is about as deep as the builder allows"""
    tri = np.zeros((n, 3, 4), np.float32)
    for i in range(n):
        s = 0.82 ** i
        c = np.array([1.0 - s, 0.15 * np.sin(i * 2.4) * s, 0.15 * np.cos(i * 2.4) * s])
        tri[i, :, :3] = c + s * 0.45 * rng.normal(size=(3, 3))
    return tri.reshape(-1, 4)


def test_deep_two_level_structure_on_the_tiled_kernel(monkeypatch):
    """k_trace_tiled_inst empties its lanes' stacks into the wave's entry pool once enough lanes are finished; the pool
    holds kCap - 120 = 712 entries and a busy lane can bring stack_entries + 1, so with deep stacks fewer lanes may still be busy at the
    switch (steal floor = 64 - 712 / (stack + 1), as the one-level kernel has it).  Deep meshes under a top level of a hundred
    overlapping instances: the tiled form (default for this launch size) against the one-ray-per-lane walk and the oracle."""
    import torch
    from optix_raytracer_amd import api
    rng = np.random.default_rng(12)
    results = []
    for env in (None, "2"):
        if env is None:
            monkeypatch.delenv("MIPT_INST_FORM", raising=False)
        else:
            monkeypatch.setenv("MIPT_INST_FORM", env)
        ctx = api.Context(0)
        rng = np.random.default_rng(12)
        # the regime the fix is for: deep enough that the floor moves (stack + 1 > 712 / 32), shallow enough for the tiled form --
        # the spikes' length is searched, so the test keeps its regime when a builder changes
        for spike in (40, 34, 46, 28, 52, 22):
            rng = np.random.default_rng(12)
            keep, gas = [], []
            for m in range(2):
                v = _spike_mesh(spike + 6 * m, rng)
                dv = torch.from_numpy(np.ascontiguousarray(v)).cuda()
                keep.append(dv)
                gas.append(ctx.accel_build([{"vertices": dv.data_ptr(), "num_vertices": v.shape[0], "stride": 16}]))
            inst = []
            for i in range(120):
                q = rng.standard_normal(4); q /= np.linalg.norm(q)
                w, x, y, z = q
                R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                              [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                              [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                xf = np.zeros((3, 4), np.float32)
                xf[:, :3] = R * rng.uniform(0.6, 1.5, 3)
                xf[:, 3] = rng.uniform(-1.0, 1.0, 3)                    # all instances overlap: a ray opens many of them
                inst.append((gas[i % 2], xf))
            ias = ctx.accel_build_instances(inst)
            info = ctx.accel_info(ias)
            if 23 <= info["stack_entries"] <= 31:
                break
        assert info["num_instances"] == 120 and 23 <= info["stack_entries"] <= 31, info
        n = 400000                                                # > 512 rays per CU: the tiled closest-hit form
        rays = np.zeros((n, 8), np.float32)
        rays[:, :3] = rng.normal(size=(n, 3)) * 4.0
        tgt = rng.uniform(-1.2, 1.2, (n, 3))
        d = tgt - rays[:, :3]
        rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
        rays[:, 3], rays[:, 7] = 1e-3, 1e16
        d_rays = torch.from_numpy(rays).cuda()
        d_hits = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
        ctx.trace_closest(ias, d_rays.data_ptr(), n, d_hits.data_ptr()); ctx.sync()
        results.append(d_hits.cpu().numpy())
        if env is None:
            nodes, order, inv, base, gas_handles = ctx.accel_export_instances(ias)
            exports = {int(h): ctx.accel_export(int(h)) for h in set(int(x) for x in gas_handles)}
            total = int(info["num_triangles"])
            osc = O.Scene({"vertices": np.zeros((total * 3, 4), np.float32), "mat_index": np.zeros(total, np.uint32),
                           "emission": np.zeros((1, 3), np.float32), "diffuse": np.ones((1, 3), np.float32)})
            osc.set_instances(nodes, order, inv, base, [exports[int(h)] for h in gas_handles])
            sub = slice(0, 60000)
            t, prim, u, v = osc.trace_closest(rays[sub], O.MODE_INST)
            h = results[0][sub]
            assert (prim != 0xffffffff).mean() > 0.3
            assert np.array_equal(h[:, 1].view(np.uint32), prim) and np.array_equal(h[:, 0].view(np.uint32), t.view(np.uint32))
        ctx.destroy()
    assert np.array_equal(results[0].view(np.uint32), results[1].view(np.uint32))


def test_single_instance_structure_runs_on_the_fast_kernels():
    """Most glTF files are ONE mesh under one node (Duck.gltf, the asset BASELINE's config 3 names), and sutil::Scene builds an instance
    structure for every scene, as the reference does.  Until round 4 a structure of one instance kept the one-ray-per-lane walk; its top
    level -- a root whose second child is empty -- now gets the entry-pool arena like any other (the empty child becomes `no child` in the
    BVH4 mirror).  Path-traced frame, counters and 0.2 M hit records (tiled and pool forms) against the oracle's two-level walk."""
    import os
    import torch
    from optix_raytracer_amd import api, scenes
    duck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "duck", "Duck.gltf")
    ctx = api.Context(0)
    flat, ias, gas, inst, keep = _build(ctx, torch, duck)
    info = ctx.accel_info(ias)
    assert info["num_instances"] == 1 and len(inst) == 1
    assert info["node_format"] == 2 and 1 <= info["stack_entries"] <= 31, info          # the arena of the fast kernels exists
    w, h, spp, depth = 320, 180, 2, 4
    launch = scenes.launch_desc(flat, w, h, spp)
    PT, st = _state(ctx, torch, flat, ias, w, h, spp, depth)
    acc, frame, stats = _render(PT, st)
    osc = _oracle_two_level(ctx, flat, ias)
    a, f, ost = osc.render(launch, mode=O.MODE_INST, max_depth=depth, threads=8)
    assert np.array_equal(acc.view(np.uint32), a.view(np.uint32)) and np.array_equal(frame, f)
    assert stats["closest_rays"] == ost["closest_rays"] and stats["shadow_rays"] == ost["shadow_rays"] > 100
    rays = osc.primary_rays(scenes.launch_desc(flat, 640, 360))
    n = rays.shape[0]
    assert n > 140000                                           # more than 512 rays per CU: the tiled two-level form
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    ctx.trace_closest(ias, d_rays.data_ptr(), n, d_hits.data_ptr()); ctx.sync()
    t, prim, u, v = osc.trace_closest(rays, O.MODE_INST)
    hg = d_hits.cpu().numpy()
    assert (prim != 0xffffffff).mean() > 0.02
    assert np.array_equal(hg[:, 1].view(np.uint32), prim) and np.array_equal(hg[:, 0].view(np.uint32), t.view(np.uint32))
    seg = rays.copy(); seg[:, 7] = 3.0
    d_seg = torch.from_numpy(seg).cuda()
    d_occ = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.trace_any(ias, d_seg.data_ptr(), n, d_occ.data_ptr()); ctx.sync()
    assert np.array_equal(d_occ.cpu().numpy().astype(np.uint8), osc.trace_any(seg, O.MODE_INST))
    ctx.destroy()
