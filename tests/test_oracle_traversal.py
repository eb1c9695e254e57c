"""Oracle self-consistency: BVH traversal (own builder) vs brute force, and structural checks.
The ray/triangle + box arithmetic is build-defined (the reference's lives in libnvoptix); what these
tests pin is that the definition is independent of the BVH in everything but exact near-ties."""
import numpy as np
import pytest

import _oracle as O
from optix_raytracer_amd import scenes


def _random_rays(n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32)
    r[:, :3], r[:, 3], r[:, 4:7], r[:, 7] = o, 0.01, d, 1e16
    return r


def test_cornell_bvh_equals_brute_force():
    sc = scenes.cornell_box()
    s = O.Scene(sc).build_bvh()
    assert s.validate() == 0
    launch = scenes.launch_desc(sc, 96, 96)
    rays = np.concatenate([s.primary_rays(launch), _random_rays(20000, 1.0, 550.0, 3)])
    t0, p0, u0, v0 = s.trace_closest(rays, O.MODE_BRUTE)
    t1, p1, u1, v1 = s.trace_closest(rays, O.MODE_BVH)
    assert np.array_equal(p0, p1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    assert np.array_equal(u0.view(np.uint32), u1.view(np.uint32)) and np.array_equal(v0.view(np.uint32), v1.view(np.uint32))
    assert (p0 != 0xffffffff).mean() > 0.8        # box open towards the camera: most rays hit
    # any-hit: bounded segments
    seg = rays.copy()
    seg[:, 7] = 300.0
    assert np.array_equal(s.trace_any(seg, O.MODE_BRUTE), s.trace_any(seg, O.MODE_BVH))


@pytest.mark.parametrize("n_tris,leaf", [(1, 4), (3, 4), (5, 2), (2000, 4), (20000, 1), (20000, 8)])
def test_soup_bvh_equals_brute_force(n_tris, leaf):
    sc = scenes.random_soup(n_tris, edge_scale=0.05 if n_tris <= 2000 else 0.01)
    s = O.Scene(sc).build_bvh(leaf_max=leaf)
    assert s.validate() == 0
    rays = _random_rays(3000, -0.2, 1.2, n_tris)
    rays[:, 4:7] = (np.array([0.5, 0.5, 0.5]) - rays[:, :3]) + np.random.default_rng(5).normal(size=(3000, 3)) * 0.3
    t0, p0, _, _ = s.trace_closest(rays, O.MODE_BRUTE)
    t1, p1, _, _ = s.trace_closest(rays, O.MODE_BVH)
    assert np.array_equal(p0, p1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    seg = rays.copy()
    seg[:, 7] = 0.3
    assert np.array_equal(s.trace_any(seg, O.MODE_BRUTE), s.trace_any(seg, O.MODE_BVH))


def test_empty_scene_and_degenerate_rays():
    sc = scenes.random_soup(0)
    s = O.Scene(sc).build_bvh()
    rays = _random_rays(10, 0, 1, 0)
    t, p, _, _ = s.trace_closest(rays)
    assert (p == 0xffffffff).all() and (t == -1).all()
    # empty interval, axis-parallel direction, zero direction: all must terminate and miss/behave
    sc = scenes.cornell_box()
    s = O.Scene(sc).build_bvh()
    r = np.array([[278, 273, 100, 5.0, 0, 0, 1, 1.0],       # tmax < tmin
                  [100, 400, 100, 0.01, 0, 0, 1, 1e16],     # axis parallel, above the blocks -> back wall z=559.2
                  [278, 273, 100, 0.01, 0, 0, 0, 1e16]],    # zero direction
                 np.float32)
    for mode in (O.MODE_BRUTE, O.MODE_BVH):
        t, p, _, _ = s.trace_closest(r, mode)
        assert p[0] == 0xffffffff and p[2] == 0xffffffff
        assert p[1] in (4, 5) and abs(t[1] - 459.2) < 1e-3


def test_tie_break_lowest_primitive():
    # two identical triangles: the hit must report primitive 0 in both modes
    v = np.array([[0, 0, 1, 0], [1, 0, 1, 0], [0, 1, 1, 0]] * 2, np.float32)
    sc = {"vertices": v, "mat_index": np.zeros(2, np.uint32), "emission": np.zeros((1, 3), np.float32),
          "diffuse": np.ones((1, 3), np.float32)}
    s = O.Scene(sc).build_bvh(leaf_max=1)
    r = np.array([[0.2, 0.2, 0, 0.01, 0, 0, 1, 1e16]], np.float32)
    for mode in (O.MODE_BRUTE, O.MODE_BVH):
        t, p, _, _ = s.trace_closest(r, mode)
        assert p[0] == 0 and t[0] == 1.0
    # hit interval is open at both ends
    r = np.array([[0.2, 0.2, 0, 1.0, 0, 0, 1, 1e16], [0.2, 0.2, 0, 0.0, 0, 0, 1, 1.0]], np.float32)
    t, p, _, _ = s.trace_closest(r, O.MODE_BVH)
    assert (p == 0xffffffff).all()


def test_depth_cap_forces_balanced_tree():
    sc = scenes.random_soup(4096)
    s = O.Scene(sc)
    s.build_bvh(leaf_max=4, max_depth=10)       # 4 * 2^10 = 4096: only a perfectly balanced tree fits
    assert s.validate() == 0 and s.depth <= 10
    with pytest.raises(RuntimeError):
        O.Scene(scenes.random_soup(4097)).build_bvh(leaf_max=4, max_depth=10)
