"""glTF loader (csrc/scene.cpp) against a synthetic scene with a numpy statement of sutil's matrix rules, against
the KATs SURVEY.md 8a17 took from the reference's own loader for SDK/data/Duck/Duck.gltf (only where that file is
present: this container), and on the GPU against the oracle."""
import os

import numpy as np
import pytest

import _gltf_fixture as F
import _oracle as O
from optix_raytracer_amd import api, scenes

DUCK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "duck", "Duck.gltf")     # data fixture: tests/golden/duck/README.md


def test_synthetic_scene_flattening_is_exact(tmp_path):
    gltf, glb, exp = F.write(tmp_path)
    for path in (gltf, glb):
        sc = api.Scene(path)
        info = sc.info()
        assert info["num_meshes"] == exp["n_meshes"] and info["num_instances"] == exp["n_instances"]
        assert info["num_cameras"] == 1 and info["num_materials"] == 2 and info["num_buffers"] == 2
        v, m = sc.flatten()
        assert info["num_triangles"] == exp["mat_index"].shape[0] == v.shape[0] // 3
        assert np.array_equal(v.view(np.uint32), exp["vertices"].view(np.uint32))
        assert np.array_equal(m, exp["mat_index"])
        cam = sc.camera()
        assert np.array_equal(cam["eye"], exp["eye"]) and np.array_equal(cam["up"], exp["up"])
        assert np.float32(cam["fovY"]) == exp["fovY"] and np.float32(cam["aspect"]) == exp["aspect"]
        # first camera looks at the scene aabb centre (Scene.cpp:687-688); the aabb comes from accessor min/max
        assert np.allclose(cam["lookat"], (info["aabb_min"] + info["aabb_max"]) * 0.5)
        w = v[:, :3]
        assert (w.min(0) >= info["aabb_min"] - 1e-5).all() and (w.max(0) <= info["aabb_max"] + 1e-5).all()
        mats = sc.materials()
        assert np.allclose(mats[0]["base_color"], (0.8, 0.1, 0.1, 1.0)) and mats[0]["metallic"] == 0.0 and mats[0]["roughness"] == 0.5
        assert np.allclose(mats[1]["base_color"], 1.0) and mats[1]["metallic"] == 1.0 and mats[1]["alpha_mode"] == 1
        assert mats[1]["alpha_cutoff"] == 0.25 and mats[1]["double_sided"]


def test_default_camera_and_errors(tmp_path):
    import json
    gltf, _, _ = F.write(tmp_path)
    doc = json.loads(gltf.read_text())
    doc["nodes"] = doc["nodes"][:4]
    doc["nodes"][0]["children"] = [1, 3]
    del doc["cameras"]
    p = tmp_path / "nocam.gltf"
    p.write_text(json.dumps(doc))
    sc = api.Scene(p)
    info, cam = sc.info(), sc.camera()
    centre = (info["aabb_min"] + info["aabb_max"]) * np.float32(0.5)
    assert info["num_cameras"] == 0 and cam["fovY"] == 45.0
    assert np.array_equal(cam["lookat"], centre)
    assert np.allclose(cam["eye"], centre + (0, 0, 1.5 * max(info["aabb_max"] - info["aabb_min"])))     # Scene.cpp:786-789
    with pytest.raises(api.MiptError):
        api.Scene(tmp_path / "missing.gltf")
    bad = tmp_path / "bad.gltf"
    bad.write_text(gltf.read_text()[:200])
    with pytest.raises(api.MiptError):
        api.Scene(bad)
    doc["accessors"][1]["count"] = 10 ** 6            # index accessor reaching past its buffer
    p2 = tmp_path / "oob.gltf"
    p2.write_text(json.dumps(doc))
    with pytest.raises(api.MiptError):
        api.Scene(p2)
    doc["accessors"][1]["count"] = 6
    doc["accessors"][1]["componentType"] = 5121       # unsigned byte indices: "component type not supported" (:98-99)
    p3 = tmp_path / "u8.gltf"
    p3.write_text(json.dumps(doc))
    with pytest.raises(api.MiptError) as e:
        api.Scene(p3)
    assert "not supported" in str(e.value)


def test_malformed_accessors_are_refused(tmp_path):
    """An accessor whose `type` is narrower than what the attribute is read as, or whose numbers are negative, huge or
    non-finite, must be refused at load -- not read past the end of its buffer in mipt_scene_flatten (found with ASan:
    a SCALAR float POSITION of count 3 over a 12-byte buffer used to load)."""
    import base64
    import json
    tri = np.array([0, 0, 0], np.float32).tobytes()                     # 12 bytes: ONE vec3, or three scalars
    def doc(acc, bv=None, extra_attr=None):
        attrs = {"POSITION": 0}
        accs = [acc]
        if extra_attr:
            attrs[extra_attr[0]] = 1
            accs.append(extra_attr[1])
        return {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
                "meshes": [{"primitives": [{"attributes": attrs}]}],
                "buffers": [{"byteLength": 12, "uri": "data:application/octet-stream;base64," + base64.b64encode(tri).decode()}],
                "bufferViews": [bv or {"buffer": 0, "byteOffset": 0, "byteLength": 12}], "accessors": accs}
    good = {"bufferView": 0, "componentType": 5126, "count": 1, "type": "VEC3"}
    ok = tmp_path / "ok.gltf"
    ok.write_text(json.dumps(doc(good)))
    assert api.Scene(ok).info()["num_triangles"] == 0
    cases = {
        "scalar_position": doc(dict(good, type="SCALAR", count=3)),
        "vec2_position": doc(dict(good, type="VEC2", count=1)),
        "count_past_end": doc(dict(good, count=2)),
        "negative_offset": doc(dict(good, byteOffset=-4)),
        "huge_offset": doc(dict(good, byteOffset=1e300)),
        "fractional_count": doc(dict(good, count=1.5)),
        "huge_count": doc(dict(good, count=2 ** 40)),
        "stride_smaller_than_element": doc(good, bv={"buffer": 0, "byteOffset": 0, "byteLength": 12, "byteStride": 4}),
        "stride_too_large": doc(good, bv={"buffer": 0, "byteOffset": 0, "byteLength": 12, "byteStride": 70000}),
        "negative_view_offset": doc(good, bv={"buffer": 0, "byteOffset": -1, "byteLength": 12}),
        "unknown_type": doc(dict(good, type="VEC5")),
        "scalar_normal_is_ignored_not_read": None,
    }
    for name, d in cases.items():
        if d is None:
            continue
        p = tmp_path / (name + ".gltf")
        p.write_text(json.dumps(d).replace("1e+300", "1e300"))
        with pytest.raises(api.MiptError):
            api.Scene(p)
    # a NORMAL / TEXCOORD / COLOR accessor of the wrong shape is treated as absent (the reference reads float VEC2 / VEC4 only,
    # Scene.cpp:502-531), never read as a wider element
    d = doc(dict(good, count=1), extra_attr=("NORMAL", {"bufferView": 0, "componentType": 5126, "count": 3, "type": "SCALAR"}))
    p = tmp_path / "scalar_normal.gltf"
    p.write_text(json.dumps(d))
    sc = api.Scene(p)
    assert sc.info()["num_triangles"] == 0


def test_duck_kats_from_the_reference_loader():
    """values printed by the reference's own Scene/Matrix/Aabb/Camera code for Duck.gltf"""
    sc = api.Scene(DUCK)
    info, cam = sc.info(), sc.camera()
    assert info["num_triangles"] == 4212 and info["num_buffers"] == 1 and info["num_images"] == 1 and info["num_materials"] == 1
    f = np.float32
    assert np.array_equal(cam["eye"], np.array([4.0011301, 4.63263988, -4.31078005], f))
    assert np.array_equal(cam["up"], np.array([-0.00425204914, 0.00783693418, 0.00452797301], f))
    assert f(cam["fovY"]) == f(37.8491669) and cam["aspect"] == 1.5
    assert np.array_equal(info["aabb_min"], np.array([-0.692984998, 0.0992936939, -0.613281965], f))
    assert np.array_equal(info["aabb_max"], np.array([0.961798966, 1.63969994, 0.539251983], f))
    assert np.array_equal(cam["lookat"], np.array([0.134406984, 0.869496822, -0.037014991], f))
    assert f(max(info["aabb_max"] - info["aabb_min"])) == f(1.65478396)
    U, V, W = api.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], f(1920) / f(1080))
    assert np.array_equal(U, np.array([-3.0930624, -0.0406331941, -2.83425117], f))
    assert np.array_equal(V, np.array([-0.885804892, 1.97587883, 0.938365459], f))
    assert np.array_equal(W, np.array([-3.86672306, -3.76314306, 4.27376509], f))
    m = sc.materials()[0]
    assert m["metallic"] == 0.0 and m["base_color_texture"] == 0 and np.allclose(m["base_color"], 1.0)
    # the oracle path-traces it (config 3 at a small size) and BVH == brute force
    scene = scenes.from_gltf(DUCK)
    osc = O.Scene(scene).build_bvh()
    launch = scenes.launch_desc(scene, 48, 27, 2)
    a0, f0, s0 = osc.render(launch, O.MODE_BVH, max_depth=8, threads=4)
    a1, f1, s1 = osc.render(dict(launch), O.MODE_BRUTE, max_depth=8, threads=8, rows=(10, 14))
    assert np.array_equal(a0[10:14].view(np.uint32), a1[10:14].view(np.uint32))
    assert s0["shadow_rays"] > 20 and (a0[..., :3] > 0).mean() > 0.01      # the duck is small in frame, lit from above


@pytest.mark.gpu
def test_gltf_scene_renders_like_the_oracle(tmp_path):
    import torch
    assert torch.cuda.is_available()
    from optix_raytracer_amd import pathtracer as PT
    gltf, _, _ = F.write(tmp_path, n_grid=24)
    scene = scenes.from_gltf(gltf)
    w, h, spp, depth = 160, 120, 2, 8
    st = PT.make_state(scene, w, h, spp, 0, max_depth=depth)
    ob = PT.OutputBuffer(w, h, st.device)
    PT.launchSubframe(ob, st)
    acc, frame, stats = st.accum.cpu().numpy(), ob.getHostPointer(), st.context.launch_stats()
    nodes, tris = st.context.accel_export(st.gas_handle)
    osc = O.Scene(scene).set_bvh(nodes, tris)
    a, f, ost = osc.render(scenes.launch_desc(scene, w, h, spp), max_depth=depth, threads=4)
    assert np.array_equal(acc.view(np.uint32), a.view(np.uint32)) and np.array_equal(frame, f)
    assert stats["closest_rays"] == ost["closest_rays"] and stats["shadow_rays"] == ost["shadow_rays"] > 0
    PT.cleanupState(st)
