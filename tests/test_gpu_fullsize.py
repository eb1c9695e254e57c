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


@pytest.mark.parametrize("builder", [4, 3])
def test_headline_frame_over_a_tree_with_spatial_splits(big, monkeypatch, builder):
    """MIPT_OPT_BUILD = 4 (what bench.py builds; 3: the same on the host): the headline scene's tree with spatial splits -- triangles that straddle a split plane
    are referenced on both sides, each reference bounding its part of the triangle conservatively -- must lose no hit: the full frame,
    the frame buffer and the ray counters at depth 1, and 40 000 pixels at depth 8, equal those over the default builder's tree bit for bit;
    the oracle walks the exported tree (more triangle records than triangles) to the same frame."""
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT2
    from optix_raytracer_amd import scenes
    PT, st, ob, scene = big
    acc, frame, stats = _render(PT, st, ob)
    monkeypatch.setenv("MIPT_BUILD", str(builder))
    st3 = PT2.make_state(scene, 1920, 1080, 1, 0, max_depth=1)
    monkeypatch.delenv("MIPT_BUILD")
    ob3 = PT2.OutputBuffer(1920, 1080, st3.device)
    acc3, frame3, stats3 = _render(PT2, st3, ob3)
    info = st3.context.accel_info(st3.gas_handle)
    assert 1_000_000 < info["num_triangles"] <= 1_500_001 and info["depth"] <= 32 and info["leaf_max"] <= 4
    assert np.array_equal(acc3.view(np.uint32), acc.view(np.uint32)) and np.array_equal(frame3, frame)
    assert stats3["closest_rays"] == stats["closest_rays"] and stats3["shadow_rays"] == stats["shadow_rays"]
    nodes, tris = st3.context.accel_export(st3.gas_handle)
    osc = O.Scene(scene).set_bvh(nodes, tris)
    a, f, ost = osc.render(scenes.launch_desc(scene, 1920, 1080, 1), max_depth=1, threads=16)
    assert np.array_equal(acc3.view(np.uint32), a.view(np.uint32)) and np.array_equal(frame3, f)
    if builder == 4:             # the device builder moves references with atomics: a second build is the same tree all the same
        PT2.buildMeshAccel(st3, scene)
        nodes2, tris2 = st3.context.accel_export(st3.gas_handle)
        assert np.array_equal(nodes.view(np.uint32), nodes2.view(np.uint32)) and np.array_equal(tris.view(np.uint32), tris2.view(np.uint32))
        del nodes2, tris2
    for s_ in (st, st3):
        s_.context.set_option(L.OPT_MAX_DEPTH, 8)
    a8, _, s8 = _render(PT, st, ob)
    b8, _, t8 = _render(PT2, st3, ob3)
    st.context.set_option(L.OPT_MAX_DEPTH, 1)
    assert np.array_equal(a8.view(np.uint32), b8.view(np.uint32)) and s8["closest_rays"] == t8["closest_rays"] and s8["shadow_rays"] == t8["shadow_rays"]
    # a refit (here: to the same vertices) gives every record the box of its WHOLE triangle: a sound tree still, the same frame
    st3.context.set_option(L.OPT_MAX_DEPTH, 1)
    st3.context.accel_update(st3.gas_handle, [{"vertices": st3.d_vertices.data_ptr(), "num_vertices": 3_000_000, "stride": 16}])
    nodes_r, tris_r = st3.context.accel_export(st3.gas_handle)
    assert np.array_equal(tris_r.view(np.uint32), tris.view(np.uint32))
    O.Scene(scene).set_bvh(nodes_r, tris_r)
    acc_r, frame_r, stats_r = _render(PT2, st3, ob3)
    assert np.array_equal(acc_r.view(np.uint32), acc.view(np.uint32)) and np.array_equal(frame_r, frame) and stats_r["shadow_rays"] == stats["shadow_rays"]
    PT2.cleanupState(st3)


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


@pytest.mark.parametrize("world", [8, 2, 3])
def test_headline_shards_equal_full_frame(big, world):
    """world 8: every shard is a pass of 259 k paths (the path kernel from the first bounce); world 2 and 3: tiled launches that
    make their own primary rays (RaygenJob) and, launched twice, walk their tiles in cost order the second time"""
    import torch
    from optix_raytracer_amd import api
    PT, st, ob, scene = big
    full, full_frame, _ = _render(PT, st, ob)
    ns = api.shard_num_samples(1920, 1080, world)
    assert ns * world >= 1920 * 1080
    gathered = torch.zeros((world, ns, 4), dtype=torch.float32, device="cuda")
    acc = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
    frame = torch.zeros((1080, 1920, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()          # cleared on torch's stream, written by launches on the state's stream (as soon as a path ends)
    for r in range(world):
        for _ in range(2):
            st.context.launch_shard(st.params, 1920, 1080, r, world, gathered[r].data_ptr(), stream=st.stream.cuda_stream)
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


def test_cost_ordered_tiles_change_nothing_but_the_order():
    """MIPT_OPT_TILE_ORDER: from the second launch on the primary closest-hit launch walks its 64-ray tiles heaviest first, in
    an order made inside the previous launch (class bytes + per-chunk histograms -> counting sort in the shade launch).  A wrong
    order (a tile twice, a tile missing) shows as wrong pixels: every launch is compared with the same launch in image order,
    with the camera moving between launches (so the order is stale), a tile count that does not fill the last sort chunk
    (1280x720: 14400 tiles = 28.125 chunks), a change of size in between, and the oracle on random pixels at the end."""
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    from optix_raytracer_amd.camera import Camera
    scene = scenes.random_soup(300_000, seed_tag=5)
    cam = scene["camera"]
    frames = {}
    for mode in (1, 0):
        st = PT.make_state(scene, 1280, 720, 1, 0, max_depth=1)
        st.context.set_option(L.OPT_TILE_ORDER, mode)
        assert st.context.get_option(L.OPT_TILE_ORDER) == mode
        ob = PT.OutputBuffer(1280, 720, st.device)
        out = []
        for k in range(5):
            eye = (cam["eye"][0] + 0.15 * k, cam["eye"][1] - 0.1 * k, cam["eye"][2])
            PT.handleCameraUpdate(st, Camera(eye, cam["lookat"], cam["up"], cam["fovY"], 1280 / 720))
            if k == 3:                            # another tile count: the tables are laid out again, the first launch is in image order
                PT.initLaunchParams(st, 1600, 900, 1)
                ob = PT.OutputBuffer(1600, 900, st.device)
                PT.handleCameraUpdate(st, Camera(eye, cam["lookat"], cam["up"], cam["fovY"], 1600 / 900))
            st.params.subframe_index = k
            PT.launchSubframe(ob, st)
            out.append((st.accum.cpu().numpy().copy(), st.context.launch_stats()))
        frames[mode] = out
        if mode == 1:
            nodes, tris = st.context.accel_export(st.gas_handle)
        PT.cleanupState(st)
    for (a1, s1), (a0, s0) in zip(frames[1], frames[0]):
        assert a1.shape == a0.shape and np.array_equal(a1.view(np.uint32), a0.view(np.uint32)) and s1 == s0
    assert not np.array_equal(frames[1][1][0], frames[1][2][0])          # the camera did move
    # the accumulation of subframes 3 and 4 at 1600x900 against the oracle
    osc = O.Scene(scene).set_bvh(nodes, tris)
    idx = np.random.default_rng(3).integers(0, 1600 * 900, 20000).astype(np.uint32)
    acc = np.zeros((idx.shape[0], 4), np.float32)
    for k in (3, 4):
        eye = (cam["eye"][0] + 0.15 * k, cam["eye"][1] - 0.1 * k, cam["eye"][2])
        sc2 = dict(scene); sc2["camera"] = dict(cam, eye=eye)
        launch = scenes.launch_desc(sc2, 1600, 900, 1, subframe=k)
        acc, _ = osc.render_pixels(launch, idx, max_depth=1, prev=acc)
    assert np.array_equal(frames[1][4][0].reshape(-1, 4)[idx].view(np.uint32), acc.view(np.uint32))


def test_one_bounce_pass_as_one_launch_gives_the_same_frame(monkeypatch):
    """MIPT_FRAME1=1 (an experiment kept in the tree: measured slower, off by default): a one-bounce pass as ONE launch of tile waves
    that make, trace and shade their paths, trace their occlusion rays and write their pixels (k_frame1) -- same arithmetic in the same
    order per path, so frames and ray counters equal the three wavefront launches' bit for bit, over several progressive subframes (the
    accumulator lerp), in cost order from the second launch on, and for a rank's shard."""
    import torch
    from optix_raytracer_amd import api
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    scene = scenes.random_soup(200_000, seed_tag=9)
    ns = api.shard_num_samples(1600, 900, 2)
    out = {}
    for mode in ("1", "2", "0"):       # (2: the tile waves trace and shade, the occlusion rays go to the wavefront form's any-hit launch)
        monkeypatch.setenv("MIPT_FRAME1", mode)
        st = PT.make_state(scene, 1600, 900, 1, 0, max_depth=1)
        ob = PT.OutputBuffer(1600, 900, st.device)
        frames = []
        for k in range(3):
            st.params.subframe_index = k
            PT.launchSubframe(ob, st)
            frames.append((st.accum.cpu().numpy().copy(), ob.getHostPointer().copy(), st.context.launch_stats()))
        shard = torch.zeros((ns, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        st.context.launch_shard(st.params, 1600, 900, 1, 2, shard.data_ptr(), stream=st.stream.cuda_stream)
        st.context.sync()
        frames.append((shard.cpu().numpy(), None, st.context.launch_stats()))
        out[mode] = frames
        PT.cleanupState(st)
    for (a1, f1, s1), (a0, f0, s0) in list(zip(out["1"], out["0"])) + list(zip(out["2"], out["0"])):
        assert np.array_equal(a1.view(np.uint32), a0.view(np.uint32))
        assert f1 is None or np.array_equal(f1, f0)
        assert s1["closest_rays"] == s0["closest_rays"] and s1["shadow_rays"] == s0["shadow_rays"] > 10000 and s1["bounces"] == s0["bounces"]


def test_cost_ordered_tiles_of_a_shard(big):
    """a rank's half of the headline frame (world 2: 1.04 M paths, the tiled kernel) walks its tiles in cost order from the
    second launch on, like a full frame; the compact shard buffer must not depend on it"""
    import torch
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import api
    PT, st, ob, scene = big
    ns = api.shard_num_samples(1920, 1080, 2)
    out = {}
    for mode in (1, 0):
        st.context.set_option(L.OPT_TILE_ORDER, mode)
        bufs = []
        for k in range(4):
            rank = 1 if k == 2 else 0                   # a change of rank in between: the order of rank 0's map is dropped
            shard = torch.zeros((ns, 4), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            st.context.launch_shard(st.params, 1920, 1080, rank, 2, shard.data_ptr(), stream=st.stream.cuda_stream)
            st.context.sync()
            bufs.append(shard.cpu().numpy())
        out[mode] = bufs
    st.context.set_option(L.OPT_TILE_ORDER, 1)
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(out[1][0], out[1][1]) and not np.array_equal(out[1][0], out[1][2])


@pytest.mark.parametrize("scene_kind,max_depth", [("soup", 0), ("cornell", 0), ("soup", 12)])
def test_path_kernel_refilling_its_waves_changes_nothing(scene_kind, max_depth, monkeypatch):
    """k_tail_pool2 as ONE set of resident waves whose lanes take new paths from the queue when theirs have ended (by default only
    over big trees; MIPT_TAIL_REFILL=2 forces it wherever the queue is longer than the resident waves hold): 786 k paths from the
    first bounce on, against the same launch without refill -- pixels, ray counters, reported depth -- and the oracle on pixels"""
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    scene = scenes.random_soup(60_000, seed_tag=9) if scene_kind == "soup" else scenes.cornell_box()
    w, h = 1024, 768
    out = {}
    for refill in ("2", "0"):
        monkeypatch.setenv("MIPT_TAIL_REFILL", refill)          # read when the context is created
        st = PT.make_state(scene, w, h, 1, 0, max_depth=max_depth)
        ob = PT.OutputBuffer(w, h, st.device)
        frames = []
        for k in range(2):
            st.params.subframe_index = k
            PT.launchSubframe(ob, st)
            frames.append((st.accum.cpu().numpy().copy(), ob.getHostPointer().copy(), st.context.launch_stats()))
        out[refill] = frames
        if refill == "2":
            nodes, tris = st.context.accel_export(st.gas_handle)
        PT.cleanupState(st)
    for (a2, f2, s2), (a0, f0, s0) in zip(out["2"], out["0"]):
        assert np.array_equal(a2.view(np.uint32), a0.view(np.uint32)) and np.array_equal(f2, f0) and s2 == s0
    osc = O.Scene(scene).set_bvh(nodes, tris)
    idx = np.random.default_rng(4).integers(0, w * h, 20000).astype(np.uint32)
    acc = np.zeros((idx.shape[0], 4), np.float32)
    for k in range(2):
        acc, _ = osc.render_pixels(scenes.launch_desc(scene, w, h, 1, subframe=k), idx, max_depth=max_depth, prev=acc)
    assert np.array_equal(out["2"][1][0].reshape(-1, 4)[idx].view(np.uint32), acc.view(np.uint32))


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


def _oracle_pixels(osc, launch, idx, max_depth, threads=16):
    """osc.render_pixels over a thread pool (the C call releases the GIL); returns (values, summed stats)"""
    from concurrent.futures import ThreadPoolExecutor
    chunks = np.array_split(idx, threads * 4)
    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(lambda c: osc.render_pixels(launch, c, max_depth=max_depth), chunks))
    return np.concatenate([r[0] for r in res]), res


def _gltf_config(tmp_path, writer, w, h, spp, depth, n_check, spp_counters, expect_tris):
    """One BASELINE glTF configuration at its specified size: the frame of ONE launch of `spp` samples, `n_check` random
    pixels against the oracle bit for bit; then a launch at spp_counters whose ray counters (and every pixel) are compared
    with an oracle render of the whole frame.  Returns the timing line."""
    import time
    import torch
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    scene = scenes.from_gltf(writer(tmp_path))
    assert expect_tris[0] <= scene["info"]["num_triangles"] <= expect_tris[1]
    st = PT.make_state(scene, w, h, spp, 0, max_depth=depth)
    ob = PT.OutputBuffer(w, h, st.device)
    PT.launchSubframe(ob, st)                                        # warm-up (allocations, caches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    PT.launchSubframe(ob, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    acc, frame, stats = st.accum.cpu().numpy(), ob.getHostPointer(), st.context.launch_stats()
    nodes, tris = st.context.accel_export(st.gas_handle)
    osc = O.Scene(scene).set_bvh(nodes, tris)
    launch = scenes.launch_desc(scene, w, h, spp)
    rng = np.random.default_rng(3)
    # half of the checked pixels are drawn among the lit ones (in these scenes most camera rays leave through the open side)
    lit = np.flatnonzero(acc.reshape(-1, 4)[:, :3].sum(1) > 0)
    idx = np.concatenate([rng.integers(0, w * h, n_check // 2), rng.choice(lit, n_check - n_check // 2)]).astype(np.uint32)
    out, _ = _oracle_pixels(osc, launch, idx, depth)
    got = acc.reshape(-1, 4)[idx]
    bad = np.flatnonzero((got.view(np.uint32) != out.view(np.uint32)).any(1))
    assert bad.size == 0, "pixel %d: gpu %s oracle %s" % (idx[bad[0]], got[bad[0]], out[bad[0]])
    assert (frame[..., 3] == 255).all() and stats["closest_rays"] >= w * h * spp
    line = "%s: %dx%d spp %d depth cap %d, %d triangles: %.1f ms per launch, %.2f Gray/s (%d closest + %d shadow rays)" % (
        scene["name"], w, h, spp, depth, scene["info"]["num_triangles"], ms, (stats["closest_rays"] + stats["shadow_rays"]) / ms / 1e6,
        stats["closest_rays"], stats["shadow_rays"])
    # batched samples (one wavefront pass for all samples): same values to rounding
    st.context.set_option(L.OPT_BATCH_SAMPLES, 1)
    if w * h * spp < 2 ** 31 // 2:
        PT.launchSubframe(ob, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        PT.launchSubframe(ob, st)
        torch.cuda.synchronize()
        line += "; batched %.1f ms" % ((time.perf_counter() - t0) * 1e3)
        accb = st.accum.cpu().numpy()
        denom = np.maximum(np.abs(acc), 1e-6)
        assert (np.abs(accb - acc) / denom).max() <= 1e-5
        sb = st.context.launch_stats()
        assert sb["closest_rays"] == stats["closest_rays"] and sb["shadow_rays"] == stats["shadow_rays"]
    st.context.set_option(L.OPT_BATCH_SAMPLES, 0)
    # ray-counter equality over the whole frame at a sample count the oracle renders in seconds
    st.params.samples_per_launch = spp_counters
    PT.launchSubframe(ob, st)
    acc2, stats2 = st.accum.cpu().numpy(), st.context.launch_stats()
    a2, _, ost2 = osc.render(scenes.launch_desc(scene, w, h, spp_counters), max_depth=depth, threads=16, want_frame=False)
    assert np.array_equal(acc2.view(np.uint32), a2.view(np.uint32))
    assert stats2["closest_rays"] == ost2["closest_rays"] and stats2["shadow_rays"] == ost2["shadow_rays"] > 0
    PT.cleanupState(st)
    print(line)
    return line


def test_config3_duck_like_gltf_1080p_spp64_depth8(tmp_path):
    """BASELINE config 3: SDK/data Duck.gltf (~4k triangles) 1920x1080 spp 64, max depth 8, NEE shadow rays, 1 GPU.  Duck.gltf
    itself stays in the reference tree; scenes.write_duck_like_gltf writes a file with its properties (4 212 triangles, u16
    indices, root scale 0.01, camera node) that goes through the same loader."""
    from optix_raytracer_amd import scenes
    _gltf_config(tmp_path, lambda d: scenes.write_duck_like_gltf(d / "duck_like.gltf"), 1920, 1080, 64, 8, 24000, 2, (4212, 4212))


def test_config3_duck_gltf_1080p_spp64_depth8(tmp_path):
    """BASELINE config 3 on the asset it names: SDK/data/Duck/Duck.gltf (4 212 triangles, u16 indices, root scale 0.01, camera node,
    base-colour texture) -- here as the data fixture tests/golden/duck/ -- at 1920x1080 spp 64, depth cap 8, NEE shadow rays."""
    import os
    duck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "duck", "Duck.gltf")
    _gltf_config(tmp_path, lambda d: duck, 1920, 1080, 64, 8, 24000, 2, (4212, 4212))


def test_config5_rooms_gltf_4k_spp256_rr_depth16(tmp_path):
    """BASELINE config 5 (its one-GPU share): a Sponza-class glTF (~250 k triangles) through the sutil::Scene-compatible loader,
    3840x2160 spp 256, Russian roulette at every bounce as the reference's loop has it, depth cap 16.  The tree holds no such
    asset; scenes.write_rooms_gltf generates one (240 k triangles, u32 indices, six materials, camera node)."""
    from optix_raytracer_amd import scenes
    _gltf_config(tmp_path, lambda d: scenes.write_rooms_gltf(d / "rooms.gltf", n_boxes=20_000), 3840, 2160, 256, 16, 20000, 1, (230_000, 260_000))
