"""Oracle path tracer: structural properties + a committed regression image.

tests/golden/oracle_cornell_32x32_spp4.npz is produced by tests/golden/make_oracle_golden.py from the
oracle itself (NOT from the reference, which cannot run here -- SURVEY.md 8c); it guards the oracle
against accidental drift, and the GPU tests compare the HIP path with the live oracle."""
import os

import numpy as np

import _oracle as O
from optix_raytracer_amd import scenes

GOLD = os.path.join(os.path.dirname(__file__), "golden", "oracle_cornell_32x32_spp4.npz")


def test_cornell_regression_image():
    sc = scenes.cornell_box()
    s = O.Scene(sc).build_bvh()
    accum, frame, st = s.render(scenes.launch_desc(sc, 32, 32, spp=4))
    g = np.load(GOLD)
    assert np.array_equal(accum.view(np.uint32), g["accum"].view(np.uint32))
    assert np.array_equal(frame, g["frame"])
    assert st["closest_rays"] == int(g["closest_rays"]) and st["shadow_rays"] == int(g["shadow_rays"])


def test_bvh_and_brute_force_render_identically():
    sc = scenes.cornell_box()
    s = O.Scene(sc).build_bvh()
    launch = scenes.launch_desc(sc, 40, 24, spp=2)
    a0, f0, s0 = s.render(launch, O.MODE_BRUTE)
    a1, f1, s1 = s.render(launch, O.MODE_BVH)
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32)) and np.array_equal(f0, f1)
    assert s0["closest_rays"] == s1["closest_rays"] and s0["shadow_rays"] == s1["shadow_rays"]


def test_render_properties():
    sc = scenes.cornell_box()
    s = O.Scene(sc).build_bvh()
    launch = scenes.launch_desc(sc, 48, 48, spp=8)
    accum, frame, st = s.render(launch, threads=4)
    a1, f1, st1 = s.render(launch, threads=1)
    assert np.array_equal(accum, a1) and st["closest_rays"] == st1["closest_rays"]     # band split is exact
    assert (accum[..., 3] == 1.0).all() and (frame[..., 3] == 255).all()
    assert np.isfinite(accum).all() and (accum[..., :3] >= 0).all()
    # the light is visible at the top centre (row 0 = bottom): emission (15,15,5) saturates the frame
    assert (frame[36:, 16:32, 0] == 255).any()
    # left of image = red wall (x=556 maps to image left because U points to -x), right = green
    left, right = accum[10:38, 1:5, :3].mean((0, 1)), accum[10:38, -5:-1, :3].mean((0, 1))
    assert left[0] > 2 * left[1] and right[1] > 2 * right[0]
    # closed scene: every continuing path hits something; about one shadow ray per diffuse hit
    assert st["closest_rays"] > 48 * 48 * 8 and 0.3 < st["shadow_rays"] / st["closest_rays"] < 1.0


def test_max_depth_and_subframes():
    sc = scenes.cornell_box()
    s = O.Scene(sc).build_bvh()
    launch = scenes.launch_desc(sc, 24, 24, spp=2)
    a_full, _, st_full = s.render(launch)
    a1, _, st1 = s.render(launch, max_depth=1)
    assert st1["closest_rays"] == 24 * 24 * 2 and st1["max_depth_seen"] == 0
    assert st_full["closest_rays"] > st1["closest_rays"]
    a8, _, st8 = s.render(launch, max_depth=8)
    assert st8["max_depth_seen"] <= 7
    # progressive accumulation: subframe 1 lerps with weight 1/2 (optixPathTracer.cu:310-316)
    l1 = dict(launch, subframe=1)
    acc = a_full.copy()
    a2, _, _ = s.render(l1, accum=acc)
    fresh, _, _ = s.render(l1, accum=np.zeros_like(acc))      # what subframe 1 alone contributes... with prev = 0
    # prev + 0.5*(cur-prev)  and  0 + 0.5*(cur-0)  =>  a2 - fresh == prev - 0.5*prev (up to rounding)
    assert np.allclose(a2[..., :3] - fresh[..., :3], 0.5 * a_full[..., :3], rtol=1e-5, atol=1e-6)


def test_render_pixels_matches_rows():
    sc = scenes.cornell_box()
    s = O.Scene(sc).build_bvh()
    launch = scenes.launch_desc(sc, 32, 16, spp=2)
    accum, _, _ = s.render(launch)
    idx = np.random.default_rng(0).permutation(32 * 16)[:100].astype(np.uint32)
    out, _ = s.render_pixels(launch, idx)
    assert np.array_equal(out, accum.reshape(-1, 4)[idx])
