"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol that
include/mipt.h declares, host helpers match the reference KATs, and the product's BVH builder emits
a tree that the oracle validates and traverses to the same answers as brute force."""
import ctypes as C
import json
import os
import re
import struct

import numpy as np
import pytest

import _oracle as O
from optix_raytracer_amd import _lib as L
from optix_raytracer_amd import api, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_kat.json")


def bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def f(u):
    return struct.unpack("<f", struct.pack("<I", u))[0]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mipt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mipt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = L.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "libmipt.so does not export %s" % name
    assert declared == {n for n, _, _ in L.SYMBOLS}, declared ^ {n for n, _, _ in L.SYMBOLS}
    assert lib.mipt_abi_version() == 4


# kept struct -> (the reference struct of tests/golden/ref_kat.json "layouts", {reference field: our field})
# The layouts come from the reference's OWN headers (oracle/ref_kat.cpp prints sizeof / alignof / offsetof of
# optixPathTracer.h:82-126, cuda/whitted.h:59-77, cuda/BufferView.h:33-49, cuda/Light.h:33-71, include/optix_types.h
# OptixInstance), not from numbers typed into this file.
KEPT_LAYOUTS = {
    "MiptParams": ("Params", None),
    "MiptParallelogramLight": ("ParallelogramLight", None),
    "MiptHitGroupData": ("HitGroupData", None),
    "MiptMissData": ("MissData", None),
    "MiptWhittedLaunchParams": ("whitted::LaunchParams", None),
    "MiptBufferView": ("BufferView", None),
    "MiptLight": ("Light", {"type": "type", "point.color": "color", "point.intensity": "intensity", "point.position": "position",
                            "point.falloff": "falloff", "ambient.color": "color"}),
    "MiptInstance": ("OptixInstance", None),
}


def test_struct_layouts_match_reference_offsets():
    """ctypes mirrors (optix_raytracer_amd/_lib.py) against the layouts the reference's headers state."""
    lay = json.load(open(GOLD))["layouts"]
    for ours, (ref, fmap) in KEPT_LAYOUTS.items():
        T, R = getattr(L, ours), lay[ref]
        assert C.sizeof(T) == R["sizeof"], (ours, C.sizeof(T), R["sizeof"])
        fmap = fmap or {k: k for k in R["fields"]}
        for rf, of in fmap.items():
            off, size = R["fields"][rf]
            d = getattr(T, of)
            assert (d.offset, d.size) == (off, size), (ours, of, (d.offset, d.size), (off, size))
        assert len(fmap) >= len([k for k in R["fields"] if "." not in k]) - (1 if ref == "Light" else 0)


def test_c_header_layouts_match_reference_offsets(tmp_path):
    """include/mipt.h itself, compiled as C99: sizeof / offsetof of the kept structs against the same reference layouts."""
    import subprocess
    lay = json.load(open(GOLD))["layouts"]
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "mipt.h"', "int main(void) {"]
    want = []
    for ours, (ref, fmap) in KEPT_LAYOUTS.items():
        R = lay[ref]
        lines.append('printf("%%zu\\n", sizeof(%s));' % ours)
        want.append(R["sizeof"])
        for rf, of in (fmap or {k: k for k in R["fields"]}).items():
            lines.append('printf("%%zu\\n", offsetof(%s, %s));' % (ours, of))
            want.append(R["fields"][rf][0])
    lines += ["return 0; }"]
    src = tmp_path / "lay.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == want


def test_reference_material_record_is_reshaped_not_kept():
    """MaterialData (cuda/MaterialData.h) holds cudaTextureObject_t handles and a Pbr / Glass / Phong union: 240 bytes inside a
    352-byte whitted::HitGroupData SBT record.  The ABI does not keep that layout (there are no texture objects on this side):
    MiptMaterialData carries the PBR branch's fields by name, MiptTextureInfo the Texture's (index instead of handle).  This test
    pins what the reference's are, so that a change of either side is seen."""
    lay = json.load(open(GOLD))["layouts"]
    assert lay["MaterialData"]["sizeof"] == 240 and lay["MaterialData::Texture"]["sizeof"] == 40
    assert lay["whitted::HitGroupData"]["sizeof"] == 352
    ref_pbr = {k.split(".", 1)[1] for k in lay["MaterialData"]["fields"] if k.startswith("pbr.")}
    ours = {n for n, _ in L.MiptMaterialData._fields_}
    assert ref_pbr <= ours, ref_pbr - ours
    assert {"alpha_mode", "alpha_cutoff", "emissive_factor", "emissive_tex", "normal_tex"} <= ours
    tex = {n for n, _ in L.MiptTextureInfo._fields_}
    assert {"texcoord", "offset", "rotation", "scale"} <= tex


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.MiptError) as e:
        api.Context(0)
    assert e.value.code == 4        # MIPT_ERROR_NO_DEVICE
    assert L.load().mipt_error_name(4) == b"MIPT_ERROR_NO_DEVICE"


def test_camera_matches_reference_kat():
    kat = json.load(open(GOLD))
    for e in kat["camera"]:
        U, V, W = api.camera_uvw([f(u) for u in e["eye"]], [f(u) for u in e["lookat"]], [f(u) for u in e["up"]],
                                 f(e["fovY"]), f(e["aspect"]), f(e["fod"]))
        assert [bits(x) for x in U] == e["U"] and [bits(x) for x in V] == e["V"] and [bits(x) for x in W] == e["W"]


def test_shard_map_matches_reference_kat():
    kat = json.load(open(GOLD))
    for e in kat["workdist"]:
        w, h, g = e["width"], e["height"], e["gpus"]
        assert api.shard_num_samples(w, h, g) == e["num_samples"]
        for gpu, smp, px, py in e["probe"]:
            assert api.shard_pixel(w, h, g, gpu, smp) == (px, py)


@pytest.mark.parametrize("w,h,world", [(1920, 1080, 8), (64, 8, 2), (100, 50, 3), (33, 7, 4)])
def test_shards_partition_the_image(w, h, world):
    ns = api.shard_num_samples(w, h, world)
    seen = np.zeros((h, w), np.int32)
    x, y = C.c_int(), C.c_int()
    Lo = O.lib()
    for r in range(world):
        for s in range(0, ns, 1 if w * h < 10000 else 97):
            px, py = api.shard_pixel(w, h, world, r, s)
            Lo.orc_work_pixel(w, h, world, r, s, C.byref(x), C.byref(y))
            assert (px, py) == (x.value, y.value)
            if px < w and py < h:
                seen[py, px] += 1
    if w * h < 10000:
        assert (seen == 1).all()        # every pixel owned by exactly one (rank, sample)
    else:
        assert seen.max() == 1


def test_random_triangles_definition():
    """centre = 3 rnd, e1/e2 = (rnd-0.5)*2*s, seed = tea<4>(i, tag); integer RNG so any host agrees"""
    n, tag, sc = 257, 0x5EED, np.float32(0.01)
    v = api.random_triangles(n, tag, float(sc))
    Lo = O.lib()
    for i in (0, 1, 100, 256):
        seed = C.c_uint32(Lo.orc_tea4(i, tag))
        r = [np.float32(Lo.orc_rnd(C.byref(seed))) for _ in range(9)]
        c = np.array(r[:3], np.float32)
        e1 = np.array([np.float32(np.float32(np.float32(x - np.float32(0.5)) * np.float32(2.0)) * sc) for x in r[3:6]], np.float32)
        e2 = np.array([np.float32(np.float32(np.float32(x - np.float32(0.5)) * np.float32(2.0)) * sc) for x in r[6:9]], np.float32)
        assert np.array_equal(v[3 * i, :3], c)
        assert np.array_equal(v[3 * i + 1, :3], c + e1) and np.array_equal(v[3 * i + 2, :3], c + e2)
        assert (v[3 * i:3 * i + 3, 3] == 0).all()
    assert v.min() > -0.02 and v.max() < 1.02


@pytest.mark.parametrize("name,n", [("cornell", 0), ("soup", 0), ("soup", 1), ("soup", 7), ("soup", 5000), ("soup", 60000)])
def test_product_bvh_is_sound_and_brute_force_equivalent(name, n):
    sc = scenes.cornell_box() if name == "cornell" else scenes.random_soup(n)
    nodes, tris, info = api.host_bvh(sc["vertices"])
    assert info["depth"] <= 32 and info["leaf_max"] == 4
    assert info["num_triangles"] == sc["vertices"].shape[0] // 3
    s = O.Scene(sc).set_bvh(nodes, tris)          # raises if the oracle's structural validation fails
    assert s.depth == info["depth"]
    rng = np.random.default_rng(n)
    lo, hi = (0.0, 556.0) if name == "cornell" else (-0.3, 1.3)
    o = rng.uniform(lo, hi, size=(4000, 3))
    tgt = rng.uniform(lo + 0.3 * (hi - lo), hi - 0.3 * (hi - lo), size=(4000, 3))
    rays = np.zeros((4000, 8), np.float32)
    rays[:, :3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 0.01, tgt - o, 1e16
    t0, p0, _, _ = s.trace_closest(rays, O.MODE_BRUTE)
    t1, p1, _, _ = s.trace_closest(rays, O.MODE_BVH)
    assert np.array_equal(p0, p1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    seg = rays.copy()
    seg[:, 7] = 0.5
    assert np.array_equal(s.trace_any(seg, O.MODE_BRUTE), s.trace_any(seg, O.MODE_BVH))


def test_product_bvh_is_deterministic():
    sc = scenes.random_soup(40000)
    a = api.host_bvh(sc["vertices"])
    b = api.host_bvh(sc["vertices"])
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@pytest.mark.parametrize("name,n", [("cornell", 0), ("soup", 5), ("soup", 7), ("soup", 5000), ("soup", 60000), ("dense", 20000)])
def test_spatial_split_bvh_is_sound_and_brute_force_equivalent(name, n, monkeypatch):
    """MIPT_OPT_BUILD = 3 (bvh_build.cpp, build_sbvh): triangles may sit in several leaves, each under boxes that bound the part of the
    triangle inside that node's region -- the oracle's validator checks exactly that, and no hit is lost or changed against brute force.
    ("dense": triangles several cells long, where the builder does split; the plain soup's tree at this size has few duplicates)"""
    if name == "cornell":
        sc = scenes.cornell_box()
    elif name == "dense":
        sc = scenes.random_soup(n, edge_scale=0.05)
    else:
        sc = scenes.random_soup(n)
    n_tris = sc["vertices"].shape[0] // 3
    nodes, tris, info = api.host_bvh(sc["vertices"], spatial=0.5)
    assert info["depth"] <= 32 and info["leaf_max"] == 4
    assert n_tris <= info["num_triangles"] <= int(n_tris * 1.5) + 1
    prims = tris.view(np.uint32)[:, 3]
    assert np.array_equal(np.unique(prims), np.arange(n_tris))          # every triangle is referenced
    if name == "dense":
        assert info["num_triangles"] > 1.1 * n_tris                      # ... and here many are referenced more than once
    s = O.Scene(sc).set_bvh(nodes, tris)
    rng = np.random.default_rng(n + 1)
    lo, hi = (0.0, 556.0) if name == "cornell" else (-0.3, 1.3)
    o = rng.uniform(lo, hi, size=(4000, 3))
    tgt = rng.uniform(lo + 0.3 * (hi - lo), hi - 0.3 * (hi - lo), size=(4000, 3))
    rays = np.zeros((4000, 8), np.float32)
    rays[:, :3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 0.01, tgt - o, 1e16
    t0, p0, _, _ = s.trace_closest(rays, O.MODE_BRUTE)
    t1, p1, _, _ = s.trace_closest(rays, O.MODE_BVH)
    assert np.array_equal(p0, p1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    seg = rays.copy()
    seg[:, 7] = 0.5
    assert np.array_equal(s.trace_any(seg, O.MODE_BRUTE), s.trace_any(seg, O.MODE_BVH))
    # the same tree whatever the number of builder threads (the reference budget is handed down the tree, not drawn from a shared counter)
    monkeypatch.setenv("MIPT_BUILD_THREADS", "1")
    nodes1, tris1, _ = api.host_bvh(sc["vertices"], spatial=0.5)
    assert np.array_equal(nodes.view(np.uint32), nodes1.view(np.uint32)) and np.array_equal(tris.view(np.uint32), tris1.view(np.uint32))


def test_oracle_validator_rejects_a_spatial_tree_that_does_not_cover_a_triangle():
    """the checker checks: in a tree with spatial splits the leaf boxes of a triangle's references must cover its corners, edge midpoints and
    centroid -- a leaf box shrunk to a speck around its triangle's centroid still overlaps the triangle's bounds, and is refused"""
    sc = scenes.random_soup(20000, edge_scale=0.05)
    nodes, tris, info = api.host_bvh(sc["vertices"], spatial=0.5)
    assert info["num_triangles"] > 20000
    O.Scene(sc).set_bvh(nodes, tris)                                    # sound as built
    refs = nodes.view(np.int32)[:, 12:14]
    prim = tris.view(np.uint32)[:, 3]
    for ni in range(nodes.shape[0]):
        for side in (0, 1):
            r = int(refs[ni, side])
            if r < 0 and ((~r) & 15) == 1:                              # a leaf of one reference
                first = ((~r) & 0xffffffff) >> 4
                v = sc["vertices"][3 * prim[first]:3 * prim[first] + 3, :3].astype(np.float64)
                c = v.mean(0)
                bad = nodes.copy()
                cols = ((0, 1), (2, 3), (8, 9)) if side == 0 else ((4, 5), (6, 7), (10, 11))
                for k, (lo, hi) in enumerate(cols):
                    bad[ni, lo], bad[ni, hi] = np.float32(c[k] - 1e-6), np.float32(c[k] + 1e-6)
                with pytest.raises(RuntimeError) as e:
                    O.Scene(sc).set_bvh(bad, tris)
                assert "-17" in str(e.value)
                return
    raise AssertionError("no single-reference leaf found")


def test_spatial_split_budget_is_kept():
    sc = scenes.random_soup(20000, edge_scale=0.05)
    n = sc["vertices"].shape[0] // 3
    for extra in (0.0, 0.1, 1.0):
        _, tris, info = api.host_bvh(sc["vertices"], spatial=extra)
        assert n <= info["num_triangles"] <= int(n * (1.0 + extra))
        if extra == 0.0:
            assert info["num_triangles"] == n
    with pytest.raises(api.MiptError):
        api.host_bvh(sc["vertices"], spatial=-1.0)


def test_header_is_plain_c(tmp_path):
    """the drop-in boundary is a C ABI: include/mipt.h compiles as C99 without warnings, and the layouts the reference's
    structs must keep have the sizes SURVEY.md 8a lists"""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "mipt.h"\n'
                   '#define SA(x) typedef char sa_##__LINE__[(x) ? 1 : -1]\n'
                   'typedef char a1[sizeof(MiptParams) == 152 ? 1 : -1];\n'
                   'typedef char a2[sizeof(MiptWhittedLaunchParams) == 128 ? 1 : -1];\n'
                   'typedef char a3[sizeof(MiptLight) == 36 ? 1 : -1];\n'
                   'typedef char a4[sizeof(MiptImguiParams) == 128 ? 1 : -1];\n'
                   'typedef char a5[sizeof(MiptImguiCamera) == 92 ? 1 : -1];\n'
                   'typedef char a6[sizeof(MiptImguiLight) == 44 ? 1 : -1];\n'
                   'typedef char a7[sizeof(MiptHitGroupData) == 32 ? 1 : -1];\n'
                   'typedef char a8[sizeof(MiptBvhNode) == 64 && sizeof(MiptBvhTri) == 48 ? 1 : -1];\n'
                   'int main(void) { return 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
