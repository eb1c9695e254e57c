"""glTF images / textures / texture transforms / per-corner attributes of the scene loader:
own PNG + DEFLATE decoder against Python's zlib-made files, sampler and KHR_texture_transform parsing
(SDK/sutil/Scene.cpp:214-265, 303-348), TEXCOORD_n / COLOR_0 flattening."""
import base64
import json

import numpy as np
import pytest

import _png_fixture as PF
from optix_raytracer_amd import api

f32 = np.float32


def _gltf_with_images(tmp_path, pngs, external=()):
    """a one-triangle scene whose textures reference the given PNG byte strings"""
    pos = np.asarray([[0, 0, 0], [1, 0, 0], [0, 1, 0]], f32)
    images = []
    for k, b in enumerate(pngs):
        if k in external:
            (tmp_path / ("img%d.png" % k)).write_bytes(b)
            images.append({"uri": "img%d.png" % k})
        else:
            images.append({"uri": "data:image/png;base64," + base64.b64encode(b).decode()})
    doc = {"asset": {"version": "2.0"},
           "buffers": [{"byteLength": pos.nbytes, "uri": "data:application/octet-stream;base64," + base64.b64encode(pos.tobytes()).decode()}],
           "bufferViews": [{"buffer": 0, "byteLength": pos.nbytes}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]}],
           "images": images, "textures": [{"source": k} for k in range(len(pngs))],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}], "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}], "scene": 0}
    p = tmp_path / "imgs.gltf"
    p.write_text(json.dumps(doc))
    return p


def test_png_decoder_matches_zlib_made_files(tmp_path):
    rng = np.random.default_rng(5)
    cases = []
    for zmode in ("dynamic", "fixed", "stored"):
        for fm in ("mix", 0, 4):
            cases.append((rng.integers(0, 256, (13, 17, 4)), 6, 8, fm, zmode, None, None))
    smooth = (np.add.outer(np.arange(40), np.arange(50))[..., None] * np.asarray([1, 2, 3]) % 256)
    cases += [
        (smooth, 2, 8, "mix", "dynamic", None, None),                                   # compressible: long matches
        (rng.integers(0, 256, (9, 11, 3)), 2, 8, "mix", "dynamic", None, (0, 7, 0, 9, 0, 11)),
        (rng.integers(0, 65536, (7, 5, 4)), 6, 16, "mix", "dynamic", None, None),
        (rng.integers(0, 65536, (6, 5, 3)), 2, 16, 3, "fixed", None, None),
        (rng.integers(0, 256, (8, 9, 2)), 4, 8, 2, "dynamic", None, None),
        (rng.integers(0, 65536, (5, 4, 2)), 4, 16, 1, "dynamic", None, None),
        (rng.integers(0, 256, (10, 10, 1)), 0, 8, "mix", "dynamic", None, (0, 100)),
        (rng.integers(0, 65536, (4, 6, 1)), 0, 16, 4, "dynamic", None, None),
        (rng.integers(0, 16, (9, 13, 1)), 0, 4, "mix", "dynamic", None, None),
        (rng.integers(0, 4, (9, 13, 1)), 0, 2, 0, "stored", None, None),
        (rng.integers(0, 2, (9, 13, 1)), 0, 1, 1, "dynamic", None, (0, 1)),
        (rng.integers(0, 200, (11, 7, 1)), 3, 8, "mix", "dynamic", rng.integers(0, 256, (200, 3)), tuple(range(0, 120, 2))),
        (rng.integers(0, 16, (11, 7, 1)), 3, 4, 0, "dynamic", rng.integers(0, 256, (16, 3)), None),
        (rng.integers(0, 2, (5, 19, 1)), 3, 1, 2, "fixed", rng.integers(0, 256, (2, 3)), (10,)),
    ]
    big = rng.integers(0, 256, (200, 300, 4))
    big[50:150] = big[0]                                                                # distances up to 32 k
    cases.append((big, 6, 8, "mix", "dynamic", None, None))
    pngs, wants = [], []
    for k, (s, ct, d, fm, zm, pal, tr) in enumerate(cases):
        b, want = PF.encode(s, ct, d, fm, zm, pal, tr, idat_split=(777 if k % 3 == 0 else 0))
        pngs.append(b); wants.append(want)
    sc = api.Scene(_gltf_with_images(tmp_path, pngs, external=(0, 5)))
    tex = sc.textures()
    assert len(tex) == len(cases) and sc.info()["num_images"] == len(cases)
    for k, (t, want) in enumerate(zip(tex, wants)):
        assert t["texels"] is not None, k
        assert (t["height"], t["width"], t["bits"]) == (want.shape[0], want.shape[1], 16 if want.dtype == np.uint16 else 8), k
        assert np.array_equal(t["texels"], want), k
        assert (t["address_s"], t["address_t"], t["filter"]) == (0, 0, 1)                # no sampler: wrap, wrap, linear


def test_undecodable_images_leave_an_empty_texture(tmp_path):
    good, _ = PF.encode(np.zeros((2, 2, 4), int), 6, 8)
    truncated = good[:-20]
    broken = bytearray(good); broken[45] ^= 0xff                                        # corrupt the deflate stream
    sc = api.Scene(_gltf_with_images(tmp_path, [good, b"\xff\xd8\xff\xe0 not a png", truncated, bytes(broken)]))
    tex = sc.textures()
    assert tex[0]["texels"] is not None
    assert all(t["texels"] is None and t["width"] == 0 for t in tex[1:])


def test_samplers_transforms_and_attributes(tmp_path):
    png, texels = PF.encode(np.arange(4 * 3 * 4).reshape(3, 4, 4) * 5, 6, 8)
    pos = np.asarray([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [2, 2, 2], [3, 2, 2], [2, 3, 2]], f32)
    uv0 = np.asarray([[0, 0], [1, 0], [0, 1], [1, 1]], f32) * f32(0.75)
    uv1 = np.asarray([[0.1, 0.2], [0.3, 0.4], [0.5, 0.6], [0.7, 0.8]], f32)
    col = np.linspace(0, 1, 12, dtype=f32).reshape(3, 4)
    idx = np.asarray([0, 1, 2, 2, 1, 3], np.uint16)
    blob = pos.tobytes() + uv0.tobytes() + uv1.tobytes() + col.tobytes() + idx.tobytes()
    o_uv0, o_uv1, o_col, o_idx = pos.nbytes, pos.nbytes + 32, pos.nbytes + 64, pos.nbytes + 64 + 48
    doc = {"asset": {"version": "2.0"},
           "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": pos.nbytes}, {"buffer": 0, "byteOffset": o_uv0, "byteLength": 32},
                           {"buffer": 0, "byteOffset": o_uv1, "byteLength": 32}, {"buffer": 0, "byteOffset": o_col, "byteLength": 48},
                           {"buffer": 0, "byteOffset": o_idx, "byteLength": 12}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]},
                         {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC2"},
                         {"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC2"},
                         {"bufferView": 3, "componentType": 5126, "count": 3, "type": "VEC4"},
                         {"bufferView": 4, "componentType": 5123, "count": 6, "type": "SCALAR"},
                         {"bufferView": 0, "byteOffset": 48, "componentType": 5126, "count": 3, "type": "VEC3", "min": [2, 2, 2], "max": [3, 3, 2]}],
           "images": [{"uri": "data:image/png;base64," + base64.b64encode(png).decode()}],
           "samplers": [{"wrapS": 33071, "wrapT": 33648, "minFilter": 9728}, {"minFilter": 9987}],
           "textures": [{"source": 0, "sampler": 0}, {"source": 0, "sampler": 1}, {"source": 0}],
           "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0, "texCoord": 1, "extensions": {"KHR_texture_transform": {
                              "offset": [0.25, -0.5], "rotation": 0.3, "scale": [2.0, 3.0]}}},
                                                    "metallicRoughnessTexture": {"index": 1}},
                          "normalTexture": {"index": 2, "texCoord": 7}, "emissiveTexture": {"index": 9}}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1, "TEXCOORD_1": 2}, "indices": 4, "material": 0},
                                      {"attributes": {"POSITION": 5, "COLOR_0": 3}}]}],
           "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}], "scene": 0}
    p = tmp_path / "attrs.gltf"
    p.write_text(json.dumps(doc))
    sc = api.Scene(p)
    tex = sc.textures()
    assert [(t["address_s"], t["address_t"], t["filter"]) for t in tex] == [(1, 2, 0), (0, 0, 1), (0, 0, 1)]
    assert all(np.array_equal(t["texels"], texels) for t in tex)
    m = sc.materials()[0]
    bt = m["base_color_tex"]
    assert bt["texture"] == 0 and bt["texcoord"] == 1 and bt["offset"] == (0.25, -0.5) and bt["scale"] == (2.0, 3.0)
    assert np.allclose(bt["rotation"], (np.sin(f32(0.3)), np.cos(f32(0.3))), rtol=1e-6)
    assert m["metallic_roughness_tex"]["texture"] == 1 and m["metallic_roughness_tex"]["rotation"] == (0.0, 1.0)
    assert m["normal_tex"]["texture"] == 2 and m["normal_tex"]["texcoord"] == 0          # set 7 does not exist: falls back to 0
    assert m["emissive_tex"]["texture"] == -1                                             # texture index out of range
    uv, has = sc.flatten_texcoords(0)
    assert has.tolist() == [1, 1, 0] and np.array_equal(uv[:6], uv0[idx]) and not uv[6:].any()
    uvb, hasb = sc.flatten_texcoords(1)
    assert hasb.tolist() == [1, 1, 0] and np.array_equal(uvb[:6], uv1[idx])
    c, hasc = sc.flatten_colors()
    assert hasc.tolist() == [0, 0, 1] and np.array_equal(c[6:], col) and (c[:6] == 1).all()


def test_jpeg_decoder_progressive(tmp_path):
    """SOF2 files (DC / AC first scans, successive-approximation refinements, EOB runs, interleaved and single-component
    scans, restart intervals) decode to exactly the pixels of a baseline file with the same coefficients
    (tests/golden/make_jpeg_progressive.py wrote both with Pillow)"""
    import pathlib
    gold = pathlib.Path(__file__).parent / "golden"
    names = "abcde"
    blobs = []
    for k in names:
        blobs += [(gold / ("jpeg_prog_%s.jpg" % k)).read_bytes(), (gold / ("jpeg_base_%s.jpg" % k)).read_bytes()]
    assert all(b"\xff\xc2" in b for b in blobs[0::2]) and not any(b"\xff\xc2" in b for b in blobs[1::2])
    sc = api.Scene(_gltf_with_images(tmp_path, blobs))
    tex = sc.textures()
    shapes = {"a": (37, 53), "b": (45, 70), "c": (24, 40), "d": (45, 70), "e": (37, 53)}
    for i, k in enumerate(names):
        prog, base = tex[2 * i], tex[2 * i + 1]
        assert prog["texels"] is not None and base["texels"] is not None, k
        assert prog["texels"].shape == shapes[k] + (4,) and prog["bits"] == 8, k
        assert np.array_equal(prog["texels"], base["texels"]), k
        assert prog["texels"][..., :3].std() > 10, k
    try:
        from PIL import Image
    except ImportError:
        return
    import io
    for i, k in enumerate(names):                 # sanity against libjpeg's own decode (other IDCT and chroma filter: loose)
        ref = np.asarray(Image.open(io.BytesIO(blobs[2 * i])).convert("RGB")).astype(int)
        assert np.abs(tex[2 * i]["texels"][..., :3].astype(int) - ref).mean() < (3.0 if k in "cd" else 12.0), k


def test_jpeg_decoder_baseline(tmp_path):
    import _jpeg_fixture as JF
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:37, 0:53]
    smooth = np.stack([(xx * 4 + yy) % 256, (yy * 6) % 256, (xx * yy) % 256], 2).astype(np.uint8)
    noisy = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    cases = [(smooth, (1, 1), 1.0, 0), (smooth, (2, 2), 1.0, 0), (smooth, (2, 1), 0.5, 3), (noisy, (2, 2), 0.3, 2), (noisy, (1, 1), 2.0, 0),
             (smooth[..., 0], (1, 1), 1.0, 0), (noisy[..., 1], (1, 1), 0.25, 5)]
    jpgs, wants = zip(*[JF.encode(img, s, qs, rst) for img, s, qs, rst in cases])
    prog, _ = JF.encode(smooth, progressive_marker=True)
    sc = api.Scene(_gltf_with_images(tmp_path, list(jpgs) + [prog], external=(1,)))
    tex = sc.textures()
    for k, (t, want) in enumerate(zip(tex, wants)):
        assert t["texels"] is not None and t["bits"] == 8, k
        assert t["texels"].shape == want.shape, k
        diff = np.abs(t["texels"].astype(int) - want.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 0.01, (k, diff.max(), (diff > 0).mean())       # float32 vs float64 rounding ties
        assert (t["texels"][..., 3] == 255).all()
    assert tex[-1]["texels"] is None                                                                   # progressive: refused, untextured
    # the decode is close to the source picture at a fine quantisation (sanity of the whole chain)
    assert np.abs(tex[0]["texels"][..., :3].astype(int) - cases[0][0].astype(int)).mean() < 15
