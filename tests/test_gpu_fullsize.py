"""Full-size checks at BASELINE.json's headline configuration (1 M random triangles, 1920x1080, spp 1):
size-independent properties + an oracle spot check of random pixels (the whole frame is also compared,
the oracle needs ~10 s for it with the exported BVH on the host cores)."""
import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    import torch
    assert torch.cuda.is_available()
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    scene = scenes.random_soup(1_000_000)
    st = PT.make_state(scene, 1920, 1080, 1, 0, max_depth=1)
    ob = PT.OutputBuffer(1920, 1080, st.device)
    yield PT, st, ob, scene
    PT.cleanupState(st)


def _render(PT, st, ob):
    PT.launchSubframe(ob, st)
    return st.accum.cpu().numpy(), ob.getHostPointer(), st.context.launch_stats()


def test_headline_frame_matches_oracle(big):
    from optix_raytracer_amd import scenes
    PT, st, ob, scene = big
    acc, frame, stats = _render(PT, st, ob)
    info = st.context.accel_info(st.gas_handle)
    assert info["num_triangles"] == 1_000_000 and info["depth"] <= 32
    nodes, tris = st.context.accel_export(st.gas_handle)
    osc = O.Scene(scene).set_bvh(nodes, tris)
    a, f, ost = osc.render(scenes.launch_desc(scene, 1920, 1080, 1), max_depth=1, threads=16)
    assert np.array_equal(acc.view(np.uint32), a.view(np.uint32))
    assert np.array_equal(frame, f)
    assert stats["closest_rays"] == ost["closest_rays"] == 1920 * 1080
    assert stats["shadow_rays"] == ost["shadow_rays"] > 100000


def test_headline_determinism_and_kernel_variants(big):
    from optix_raytracer_amd import _lib as L
    PT, st, ob, scene = big
    a0, f0, s0 = _render(PT, st, ob)
    a1, f1, s1 = _render(PT, st, ob)
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32)) and s0 == s1          # idempotent relaunch
    for variant in (0, 2):                      # the one-thread-per-ray kernel, streaming waves with subtree stealing
        st.context.set_option(L.OPT_PERSISTENT, variant)
        a2, f2, s2 = _render(PT, st, ob)
        assert np.array_equal(a0.view(np.uint32), a2.view(np.uint32)) and np.array_equal(f0, f2), variant
    st.context.set_option(L.OPT_PERSISTENT, 3)
    opts = (L.OPT_TRACE_GRID, L.OPT_REFILL_MIN, L.OPT_STEAL_MIN, L.OPT_BURST)
    defaults = [st.context.get_option(o) for o in opts]
    # incl. stealing on every idle lane / never, single-step and long bursts
    for vals in ((4, 1, 1, 0x0101), (20, 32, 65, 0x4040), (32, 64, 16, 0x0120), (8, 8, 40, 0x1002)):
        for o, v in zip(opts, vals):
            st.context.set_option(o, v)
        a3, f3, _ = _render(PT, st, ob)
        assert np.array_equal(a0.view(np.uint32), a3.view(np.uint32)) and np.array_equal(f0, f3), vals
    for o, v in zip(opts, defaults):
        st.context.set_option(o, v)


def test_headline_shards_equal_full_frame(big):
    import torch
    from optix_raytracer_amd import api
    PT, st, ob, scene = big
    full, full_frame, _ = _render(PT, st, ob)
    world = 8
    ns = api.shard_num_samples(1920, 1080, world)
    assert ns * world == 1920 * 1080
    gathered = torch.zeros((world, ns, 4), dtype=torch.float32, device="cuda")
    for r in range(world):
        st.context.launch_shard(st.params, 1920, 1080, r, world, gathered[r].data_ptr(), stream=st.stream.cuda_stream)
    acc = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
    frame = torch.zeros((1080, 1920, 4), dtype=torch.uint8, device="cuda")
    st.context.unpermute(gathered.data_ptr(), world, 1920, 1080, acc.data_ptr(), frame.data_ptr(), stream=st.stream.cuda_stream)
    st.context.sync()
    assert np.array_equal(acc.cpu().numpy().view(np.uint32), full.view(np.uint32))
    assert np.array_equal(frame.cpu().numpy(), full_frame)


def test_headline_depth8_spot_check(big):
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import scenes
    PT, st, ob, scene = big
    st.context.set_option(L.OPT_MAX_DEPTH, 8)
    acc, frame, stats = _render(PT, st, ob)
    st.context.set_option(L.OPT_MAX_DEPTH, 1)
    nodes, tris = st.context.accel_export(st.gas_handle)
    osc = O.Scene(scene).set_bvh(nodes, tris)
    idx = np.random.default_rng(1).integers(0, 1920 * 1080, 40000).astype(np.uint32)
    out, _ = osc.render_pixels(scenes.launch_desc(scene, 1920, 1080, 1), idx, max_depth=8)
    assert np.array_equal(acc.reshape(-1, 4)[idx].view(np.uint32), out.view(np.uint32))
    assert stats["closest_rays"] > 1920 * 1080


def test_config2_cornell_1080p_spp16():
    """BASELINE config 2: Cornell box 1920x1080 spp 16, one launch, unbounded (Russian roulette) depth.
    33 M paths: the oracle checks 30 k random pixels bit for bit and the ray counters of the whole frame
    are compared with an oracle render of every 8th row band."""
    import torch
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    scene = scenes.cornell_box()
    w, h, spp = 1920, 1080, 16
    st = PT.make_state(scene, w, h, spp, 0, max_depth=0)
    ob = PT.OutputBuffer(w, h, st.device)
    PT.launchSubframe(ob, st)
    acc = st.accum.cpu().numpy()
    frame = ob.getHostPointer()
    stats = st.context.launch_stats()
    nodes, tris = st.context.accel_export(st.gas_handle)
    osc = O.Scene(scene).set_bvh(nodes, tris)
    launch = scenes.launch_desc(scene, w, h, spp)
    idx = np.random.default_rng(2).integers(0, w * h, 30000).astype(np.uint32)
    out, _ = osc.render_pixels(launch, idx)
    got = acc.reshape(-1, 4)[idx]
    assert np.array_equal(got.view(np.uint32), out.view(np.uint32))
    assert (frame[..., 3] == 255).all()
    # one primary ray per pixel and sample plus the Russian-roulette survivors; every sample runs its own bounce loop
    assert stats["closest_rays"] > 1.5 * w * h * spp and stats["bounces"] >= 16 * 10
    PT.cleanupState(st)
