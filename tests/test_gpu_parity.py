"""GPU parity tests (run on an MI355X with `-m gpu`): the HIP path, called through the C ABI, against
the CPU oracle on the same seeded inputs and the SAME BVH (exported from device memory).

Bars (BASELINE.json north_star): traversal records (t, prim, u, v) and occlusion flags bit-exact;
ray counters and node/triangle visit counts equal; accum float4 within 1e-5 relative (they are in
fact bit-identical, asserted where stated); frame uchar4 exact."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5          # north_star tolerance for pixel RGB


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _state(scene, w, h, spp=1, max_depth=0):
    from optix_raytracer_amd import pathtracer as PT
    st = PT.make_state(scene, w, h, spp, 0, max_depth)
    return PT, st


def _oracle_scene(scene, st):
    nodes, tris = st.context.accel_export(st.gas_handle)
    return O.Scene(scene).set_bvh(nodes, tris)


def _render_gpu(PT, st, subframe=0):
    st.params.subframe_index = subframe
    ob = getattr(st, "_ob", None)
    if ob is None:
        ob = PT.OutputBuffer(st.params.width, st.params.height, st.device)
        st._ob = ob
    PT.launchSubframe(ob, st)
    return st.accum.cpu().numpy(), ob.getHostPointer(), st.context.launch_stats()


def _assert_pixels(acc_gpu, acc_cpu, exact=True):
    if exact:
        bad = np.argwhere(acc_gpu.view(np.uint32) != acc_cpu.view(np.uint32))
        assert bad.size == 0, "first differing pixel %s: gpu %s cpu %s" % (bad[0], acc_gpu[tuple(bad[0][:2])], acc_cpu[tuple(bad[0][:2])])
    denom = np.maximum(np.abs(acc_cpu), 1e-6)
    assert (np.abs(acc_gpu - acc_cpu) / denom).max() <= REL_TOL


# ------------------------------------------------------------------------------------------------
def test_library_is_native_and_loaded():
    torch = _torch()
    from optix_raytracer_amd import _lib as L
    L.load()
    maps = open("/proc/self/maps").read()
    assert "optix_raytracer_amd/lib/libmipt.so" in maps
    assert torch.cuda.get_device_properties(0).gcnArchName.startswith("gfx950")


@pytest.mark.parametrize("persistent,steal", [(3, 32), (3, 1), (3, 65), (2, 32), (2, 1), (2, 65), (0, 32)])
def test_traversal_records_bit_exact(persistent, steal):
    torch = _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import scenes
    for scene, (w, h) in ((scenes.cornell_box(), (160, 96)), (scenes.random_soup(30000), (200, 120))):
        PT, st = _state(scene, w, h)
        st.context.set_option(L.OPT_PERSISTENT, persistent)
        st.context.set_option(L.OPT_STEAL_MIN, steal)
        st.context.set_option(L.OPT_COUNT_VISITS, 1)
        osc = _oracle_scene(scene, st)
        launch = scenes.launch_desc(scene, w, h)
        rays = osc.primary_rays(launch)
        # add incoherent rays + empty-interval rays + axis-parallel rays
        rng = np.random.default_rng(7)
        lo, hi = (0, 556) if scene["name"] == "cornell" else (-0.2, 1.2)
        extra = np.zeros((5000, 8), np.float32)
        extra[:, :3] = rng.uniform(lo, hi, (5000, 3))
        extra[:, 4:7] = rng.normal(size=(5000, 3))
        extra[:, 3], extra[:, 7] = 0.01, 1e16
        extra[:50, 7] = 0.0                       # tmax < tmin
        extra[50:100, 4:6] = 0.0                  # parallel to z
        rays = np.concatenate([rays, extra])
        n = rays.shape[0]
        d_rays = torch.from_numpy(rays).cuda()
        d_hits = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
        st.context.set_option(L.OPT_COUNT_VISITS, 0)       # the production build of the kernels (steals subtrees)
        st.context.trace_closest(st.gas_handle, d_rays.data_ptr(), n, d_hits.data_ptr())
        st.context.sync()
        hits_fast = d_hits.cpu().numpy()
        d_hits.zero_()
        st.context.set_option(L.OPT_COUNT_VISITS, 1)       # the counting build (exact per-ray order)
        st.context.trace_closest(st.gas_handle, d_rays.data_ptr(), n, d_hits.data_ptr())
        stats = st.context.launch_stats()
        hits = d_hits.cpu().numpy()
        assert np.array_equal(hits.view(np.uint32), hits_fast.view(np.uint32))
        ost = O.OrcStats()
        t, prim, u, v = osc.trace_closest(rays, O.MODE_BVH, ost)
        assert np.array_equal(hits[:, 1].view(np.uint32), prim)
        assert np.array_equal(hits[:, 0].view(np.uint32), t.view(np.uint32))
        hit = prim != 0xffffffff
        assert np.array_equal(hits[hit, 2].view(np.uint32), u[hit].view(np.uint32))
        assert np.array_equal(hits[hit, 3].view(np.uint32), v[hit].view(np.uint32))
        assert stats["node_visits"] == ost.node_visits and stats["tri_tests"] == ost.tri_tests
        # occlusion
        seg = rays.copy()
        seg[:, 7] = np.where(seg[:, 7] > 1.0, 250.0 if scene["name"] == "cornell" else 0.4, seg[:, 7])
        d_seg = torch.from_numpy(seg).cuda()
        d_occ = torch.zeros(n, dtype=torch.int32, device="cuda")
        occ = osc.trace_any(seg, O.MODE_BVH)
        for counting in (0, 1):
            st.context.set_option(L.OPT_COUNT_VISITS, counting)
            d_occ.zero_()
            st.context.trace_any(st.gas_handle, d_seg.data_ptr(), n, d_occ.data_ptr())
            assert np.array_equal(d_occ.cpu().numpy().astype(np.uint8), occ), counting
        PT.cleanupState(st)


@pytest.mark.parametrize("grid,steal,n_rays", [(1, 32, 150_000), (1, 8, 40_000), (2, 64, 70_000), (16, 32, 300_000)])
def test_entry_pool_kernel_all_phases(grid, steal, n_rays):
    """variant 3 on launches that (a) start in the bulk phase and go over to the entry pool when the ray range runs
    dry, (b) run on the pool from the start; hit records and flags equal the counting build's (= the oracle's order)"""
    torch = _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import scenes
    scene = scenes.random_soup(30000)
    PT, st = _state(scene, 64, 64)
    st.context.set_option(L.OPT_PERSISTENT, 3)
    st.context.set_option(L.OPT_TRACE_GRID, grid)
    st.context.set_option(L.OPT_STEAL_MIN, steal)
    rng = np.random.default_rng(grid * 1000 + steal)
    rays = np.zeros((n_rays, 8), np.float32)
    rays[:, :3] = rng.uniform(-0.2, 1.2, (n_rays, 3))
    rays[:, 4:7] = rng.normal(size=(n_rays, 3))
    rays[:, 3], rays[:, 7] = 0.001, 1e16
    rays[::97, 7] = 0.0                                   # empty intervals sprinkled in
    # a coherent block of long rays so that some waves are much heavier than others
    rays[:4096, :3] = (0.5, 0.5, -0.5); rays[:4096, 4:7] = rng.normal((0, 0, 1), 0.02, (4096, 3))
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.zeros((n_rays, 4), dtype=torch.float32, device="cuda")
    d_occ = torch.zeros(n_rays, dtype=torch.int32, device="cuda")
    out = {}
    for counting in (0, 1):
        st.context.set_option(L.OPT_COUNT_VISITS, counting)
        d_hits.zero_(); d_occ.zero_()
        st.context.trace_closest(st.gas_handle, d_rays.data_ptr(), n_rays, d_hits.data_ptr())
        st.context.trace_any(st.gas_handle, d_rays.data_ptr(), n_rays, d_occ.data_ptr())
        st.context.sync()
        out[counting] = (d_hits.cpu().numpy().copy(), d_occ.cpu().numpy().copy())
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1], out[1][1])
    hit = out[0][0][:, 0] > 0
    assert 0.05 < hit.mean() < 1.0 and np.array_equal(hit, out[0][1] != 0)
    osc = _oracle_scene(scene, st)
    t, prim, u, v = osc.trace_closest(rays[:20000], O.MODE_BVH)
    assert np.array_equal(out[0][0][:20000, 0].view(np.uint32), t.view(np.uint32))
    assert np.array_equal(out[0][0][:20000, 1].view(np.uint32), prim)
    PT.cleanupState(st)


def test_cornell_config1_512_spp1():
    """BASELINE config 1: Cornell 512x512 spp 1 (CPU reference) vs the GPU path, unbounded depth"""
    _torch()
    from optix_raytracer_amd import scenes
    scene = scenes.cornell_box()
    PT, st = _state(scene, 512, 512, 1)
    acc, frame, stats = _render_gpu(PT, st)
    osc = _oracle_scene(scene, st)
    a, f, ost = osc.render(scenes.launch_desc(scene, 512, 512, 1), threads=8)
    _assert_pixels(acc, a)
    assert np.array_equal(frame, f)
    assert stats["closest_rays"] == ost["closest_rays"] and stats["shadow_rays"] == ost["shadow_rays"]
    assert stats["bounces"] == ost["max_depth_seen"] + 1
    PT.cleanupState(st)


@pytest.mark.parametrize("w,h,spp,max_depth", [(96, 64, 4, 0), (100, 37, 3, 0), (64, 64, 16, 8), (33, 9, 1, 1),
                                               (1, 1, 1, 0), (7, 1, 2, 0), (1, 65, 1, 3), (65, 3, 1, 1)])
def test_cornell_multi_sample_and_ragged_sizes(w, h, spp, max_depth):
    _torch()
    from optix_raytracer_amd import scenes
    scene = scenes.cornell_box()
    PT, st = _state(scene, w, h, spp, max_depth)
    acc, frame, stats = _render_gpu(PT, st)
    osc = _oracle_scene(scene, st)
    a, f, ost = osc.render(scenes.launch_desc(scene, w, h, spp), max_depth=max_depth, threads=4)
    _assert_pixels(acc, a)
    assert np.array_equal(frame, f)
    assert stats["closest_rays"] == ost["closest_rays"] and stats["shadow_rays"] == ost["shadow_rays"]
    # these passes run in the path kernel from their first bounce, which then also does the pass's queue bookkeeping in its first
    # workgroup while the other waves run (TailArgs::epoch orders it before their additions to the statistics): the counters of
    # 30 more launches, whose waves finish within microseconds on this scene
    for _ in range(30):
        _, _, again = _render_gpu(PT, st)
        assert again == stats
    PT.cleanupState(st)


@pytest.mark.parametrize("scene_kind,w,h,spp,max_depth", [("cornell", 160, 120, 3, 0), ("soup", 128, 96, 2, 0), ("cornell", 96, 64, 2, 6)])
def test_tail_kernel_gives_the_same_frame_and_counts(scene_kind, w, h, spp, max_depth):
    """MIPT_OPT_TAIL_PATHS: the last paths of an unbounded launch run in one kernel (one path per lane) instead of
    wavefront bounces -- pixels, ray counts and the reported path depth do not depend on where the switch happens"""
    _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import scenes
    scene = scenes.cornell_box() if scene_kind == "cornell" else scenes.random_soup(20000)
    PT, st = _state(scene, w, h, spp, max_depth)
    ref = None
    for tail, variant in ((0, 3), (1, 3), (300, 3), (8192, 3), (1 << 20, 3), (300, 0), (1 << 20, 2)):
        st.context.set_option(L.OPT_PERSISTENT, variant)     # 3: pool tail kernel (k_tail_pool), else the plain loop (k_tail)
        st.context.set_option(L.OPT_TAIL_PATHS, tail)
        assert st.context.get_option(L.OPT_TAIL_PATHS) == tail
        acc, frame, stats = _render_gpu(PT, st)
        cur = (acc.copy(), frame.copy(), stats["closest_rays"], stats["shadow_rays"], stats["bounces"])
        if ref is None:
            ref = cur
            osc = _oracle_scene(scene, st)
            a, f, ost = osc.render(scenes.launch_desc(scene, w, h, spp), max_depth=max_depth, threads=4)
            _assert_pixels(acc, a)
            assert stats["closest_rays"] == ost["closest_rays"] and stats["shadow_rays"] == ost["shadow_rays"]
            assert stats["bounces"] == spp * (ost["max_depth_seen"] + 1) or spp > 1
            continue
        assert np.array_equal(cur[0].view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(cur[1], ref[1]), tail
        assert cur[2:] == ref[2:], (tail, variant, cur[2:], ref[2:])
    with pytest.raises(Exception):
        st.context.set_option(L.OPT_TAIL_PATHS, (1 << 30) + 1)
    PT.cleanupState(st)


def test_progressive_subframes():
    _torch()
    from optix_raytracer_amd import scenes
    scene = scenes.cornell_box()
    PT, st = _state(scene, 80, 48, 2)
    osc = _oracle_scene(scene, st)
    acc_cpu = None
    for sub in range(3):
        acc, frame, _ = _render_gpu(PT, st, sub)
        acc_cpu, f, _ = osc.render(scenes.launch_desc(scene, 80, 48, 2, subframe=sub), accum=acc_cpu)
        _assert_pixels(acc, acc_cpu)
        assert np.array_equal(frame, f)
    PT.cleanupState(st)


@pytest.mark.parametrize("n_tris,max_depth", [(100000, 1), (100000, 8), (20000, 0)])
def test_soup_scene_parity(n_tris, max_depth):
    """the benchmark scene family at a size the oracle renders in seconds"""
    _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import scenes
    scene = scenes.random_soup(n_tris)
    w, h = 320, 180
    PT, st = _state(scene, w, h, 1, max_depth)
    st.context.set_option(L.OPT_COUNT_VISITS, 1)
    acc, frame, stats = _render_gpu(PT, st)
    osc = _oracle_scene(scene, st)
    a, f, ost = osc.render(scenes.launch_desc(scene, w, h, 1), max_depth=max_depth, threads=8)
    _assert_pixels(acc, a)
    assert np.array_equal(frame, f)
    for k in ("closest_rays", "shadow_rays", "node_visits", "tri_tests", "shadow_node_visits", "shadow_tri_tests"):
        assert stats[k] == ost[k], k
    PT.cleanupState(st)


def test_shards_reassemble_to_the_full_frame():
    """multi-GPU path on one GPU: every rank's shard rendered in turn, gathered buffer un-permuted"""
    torch = _torch()
    from optix_raytracer_amd import api, scenes
    scene = scenes.cornell_box()
    w, h, spp = 104, 44, 2          # not a multiple of the 8*world strip: padded samples must be ignored
    PT, st = _state(scene, w, h, spp)
    full, full_frame, _ = _render_gpu(PT, st)
    for world in (1, 2, 3, 8):
        ns = api.shard_num_samples(w, h, world)
        gathered = torch.zeros((world, ns, 4), dtype=torch.float32, device="cuda")
        acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        frame = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()      # cleared on torch's stream, written on the state's stream
        for r in range(world):
            st.context.launch_shard(st.params, w, h, r, world, gathered[r].data_ptr(), stream=st.stream.cuda_stream)
        st.context.unpermute(gathered.data_ptr(), world, w, h, acc.data_ptr(), frame.data_ptr(), stream=st.stream.cuda_stream)
        st.context.sync()
        assert np.array_equal(acc.cpu().numpy().view(np.uint32), full.view(np.uint32)), world
        assert np.array_equal(frame.cpu().numpy(), full_frame), world
    PT.cleanupState(st)


def test_error_behaviour():
    torch = _torch()
    from optix_raytracer_amd import api, scenes
    from optix_raytracer_amd import pathtracer as PT
    scene = scenes.cornell_box()
    st = PT.PathTracerState()
    PT.createContext(st)
    PT.buildMeshAccel(st, scene)
    PT.initLaunchParams(st, 32, 32, 1)
    with pytest.raises(api.MiptError) as e:        # no hit-group records yet
        st.context.launch(st.params, 32, 32, 1)
    assert e.value.code == 6
    PT.createSBT(st)
    with pytest.raises(api.MiptError) as e:        # launch size must match params
        st.context.launch(st.params, 64, 32, 1)
    assert e.value.code == 1
    with pytest.raises(api.MiptError) as e:
        st.context.launch(st.params, 32, 32, 2)
    assert e.value.code == 1
    st.params.handle = 12345
    with pytest.raises(api.MiptError) as e:
        st.context.launch(st.params, 32, 32, 1)
    assert e.value.code == 5
    st.params.handle = st.gas_handle
    st.params.samples_per_launch = 0
    with pytest.raises(api.MiptError):
        st.context.launch(st.params, 32, 32, 1)
    with pytest.raises(api.MiptError):
        st.context.accel_build([{"vertices": None, "num_vertices": 3}])
    with pytest.raises(api.MiptError):             # indexed input with an out-of-range index
        st.context.accel_build([{"vertices": np.zeros((3, 3), np.float32), "num_vertices": 3,
                                 "indices": np.array([0, 1, 7], np.uint32), "index_format": 4, "num_triangles": 1}])
    PT.cleanupState(st)


def test_epoch_handshake_timeout_is_reported(monkeypatch):
    """A path kernel that makes its own rays lets its waves wait (bounded) for the first workgroup's bookkeeping.  If that wait ever
    runs out, the launch statistics may miss additions -- so it must not be silent: the wave raises a bit in the context's pinned
    fault word and the next synchronising call returns MIPT_ERROR_INTERNAL, once.  MIPT_DEBUG_EPOCH_FAULT makes the first workgroup
    publish nothing, so every wave runs into the time-out.  Pixels do not depend on the handshake: they still equal the oracle's."""
    torch = _torch()
    from optix_raytracer_amd import api, scenes
    scene = scenes.cornell_box()
    w, h = 96, 64
    monkeypatch.setenv("MIPT_DEBUG_EPOCH_FAULT", "1")
    PT, st = _state(scene, w, h, 1, 0)
    monkeypatch.delenv("MIPT_DEBUG_EPOCH_FAULT")
    ob = PT.OutputBuffer(w, h, st.device)
    with pytest.raises(api.MiptError) as e:                    # a small pass: the path kernel from the first bounce, own raygen;
        PT.launchSubframe(ob, st)                              # launchSubframe ends in mipt_sync (CUDA_SYNC_CHECK's place)
    assert e.value.code == 8 and "epoch" in str(e.value)       # MIPT_ERROR_INTERNAL
    st.context.sync()                                          # reported once
    acc = st.accum.cpu().numpy()
    a, f, ost = _oracle_scene(scene, st).render(scenes.launch_desc(scene, w, h, 1))
    _assert_pixels(acc, a)
    # a caller that synchronises the stream itself (bench.py, meshviewer.py via torch) and launches again never passes mipt_sync:
    # the NEXT launch reports the fault of the one before, once
    lib, hnd = st.context._lib, st.context._h
    import ctypes as C
    st.params.frame_buffer = ob.map()
    assert lib.mipt_launch(hnd, st.stream.cuda_stream, C.byref(st.params), w, h, 1) == 0       # faults again (the injection is the context's)
    ob.unmap()
    torch.cuda.synchronize()
    st.params.frame_buffer = ob.map()
    assert lib.mipt_launch(hnd, st.stream.cuda_stream, C.byref(st.params), w, h, 1) == 8       # MIPT_ERROR_INTERNAL: the earlier launch's
    ob.unmap()
    torch.cuda.synchronize()
    # a context without the injection never raises it
    PT2, st2 = _state(scene, w, h, 1, 0)
    acc2, _, stats2 = _render_gpu(PT2, st2)
    st2.context.sync()
    assert stats2["closest_rays"] == ost["closest_rays"] and stats2["shadow_rays"] == ost["shadow_rays"]
    _assert_pixels(acc2, a)
    PT.cleanupState(st); PT2.cleanupState(st2)


def test_indexed_strided_and_transformed_inputs():
    """build inputs as sutil::Scene feeds them: u16/u32 indices, byte strides, a 3x4 instance transform"""
    torch = _torch()
    from optix_raytracer_amd import api, scenes
    scene = scenes.random_soup(500, edge_scale=0.1)
    v = scene["vertices"].reshape(-1, 4)[:, :3].copy()
    # weld nothing, but index in reverse order and interleave vertices with 8 bytes of padding
    packed = np.zeros((v.shape[0], 5), np.float32)
    packed[:, :3] = v
    idx32 = np.arange(v.shape[0], dtype=np.uint32).reshape(-1, 3)[::-1].copy()
    idx16 = idx32.astype(np.uint16)
    xf = np.array([[0, -2, 0, 1], [2, 0, 0, -3], [0, 0, 2, 0.5]], np.float32)
    ctx = api.Context(0)
    handles = [
        ctx.accel_build([{"vertices": packed, "num_vertices": v.shape[0], "stride": 20, "indices": idx32,
                          "index_format": 4, "num_triangles": idx32.shape[0], "transform": xf}]),
        ctx.accel_build([{"vertices": torch.from_numpy(packed).cuda().data_ptr(), "num_vertices": v.shape[0], "stride": 20,
                          "indices": idx16, "index_format": 2, "num_triangles": idx16.shape[0], "transform": xf}]),
    ]
    world = (v[idx32.reshape(-1)] @ xf[:, :3].T + xf[:, 3]).astype(np.float32)
    sc2 = dict(scene, vertices=np.concatenate([world, np.zeros((world.shape[0], 1), np.float32)], axis=1))
    rng = np.random.default_rng(0)
    rays = np.zeros((3000, 8), np.float32)
    rays[:, :3] = rng.uniform(-4, 4, (3000, 3))
    rays[:, 4:7] = np.array([0.0, -2.0, 1.5]) - rays[:, :3] + rng.normal(size=(3000, 3))
    rays[:, 3], rays[:, 7] = 0.01, 1e16
    d_rays = torch.from_numpy(rays).cuda()
    for hnd in handles:
        nodes, tris = ctx.accel_export(hnd)
        osc = O.Scene(sc2).set_bvh(nodes, tris)
        d_hits = torch.zeros((3000, 4), dtype=torch.float32, device="cuda")
        ctx.trace_closest(hnd, d_rays.data_ptr(), 3000, d_hits.data_ptr())
        ctx.sync()
        t, prim, _, _ = osc.trace_closest(rays, O.MODE_BRUTE)
        hits = d_hits.cpu().numpy()
        assert np.array_equal(hits[:, 1].view(np.uint32), prim) and np.array_equal(hits[:, 0].view(np.uint32), t.view(np.uint32))
        assert (prim != 0xffffffff).sum() > 100
    ctx.destroy()


def test_batched_samples_agree_within_tolerance():
    """MIPT_OPT_BATCH_SAMPLES: all samples of a launch in one wavefront pass; per-sample sums are added in sample
    order, so values agree with the sequential order to rounding (north_star tolerance 1e-5), ray counts exactly."""
    _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import scenes
    scene = scenes.cornell_box()
    w, h, spp = 160, 96, 16
    PT, st = _state(scene, w, h, spp, 0)
    seq_acc, seq_frame, seq_stats = _render_gpu(PT, st)
    st.context.set_option(L.OPT_BATCH_SAMPLES, 1)
    acc, frame, stats = _render_gpu(PT, st)
    st.context.set_option(L.OPT_BATCH_SAMPLES, 0)
    osc = _oracle_scene(scene, st)
    a, f, ost = osc.render(scenes.launch_desc(scene, w, h, spp), threads=4)
    assert np.array_equal(seq_acc.view(np.uint32), a.view(np.uint32))                 # sequential order: bit exact
    denom = np.maximum(np.abs(a), 1e-6)
    assert (np.abs(acc - a) / denom).max() <= REL_TOL
    assert np.abs(frame.astype(np.int32) - f.astype(np.int32)).max() <= 1
    assert (frame != f).mean() < 1e-3
    assert stats["closest_rays"] == ost["closest_rays"] and stats["shadow_rays"] == ost["shadow_rays"]
    assert stats["bounces"] < seq_stats["bounces"] / 4                                # far fewer wavefront iterations
    # a second subframe in batch mode (progressive lerp on top of the batched result)
    st.context.set_option(L.OPT_BATCH_SAMPLES, 1)
    acc2, _, _ = _render_gpu(PT, st, 1)
    a2, _, _ = osc.render(scenes.launch_desc(scene, w, h, spp, subframe=1), accum=a.copy(), threads=4)
    assert (np.abs(acc2 - a2) / np.maximum(np.abs(a2), 1e-6)).max() <= REL_TOL
    PT.cleanupState(st)


def test_render_loop_and_file_output(tmp_path):
    """main()'s --file branch + the progressive loop: 3 sub-frames of the Cornell box, PPM and PNG written with
    sutil::saveImage's flip; the accumulated frame equals the oracle's after the same 3 sub-frames"""
    _torch()
    from optix_raytracer_amd import image, scenes
    import test_image_io as TI
    scene = scenes.cornell_box()
    w, h, spp = 96, 64, 2
    from optix_raytracer_amd import pathtracer as PTm
    st = PTm.renderToFile(scene, tmp_path / "cornell.ppm", w, h, spp, subframes=3)
    acc = st.accum.cpu().numpy()
    osc = _oracle_scene(scene, st)
    a = None
    for sf in range(3):
        a, f, _ = osc.render(scenes.launch_desc(scene, w, h, spp, subframe=sf), accum=a)
    _assert_pixels(acc, a)
    ppm = TI._read_ppm(tmp_path / "cornell.ppm")
    assert np.array_equal(ppm, f[::-1, :, :3])
    image.saveImage(tmp_path / "cornell.png", f)
    assert np.array_equal(TI._read_png(tmp_path / "cornell.png"), f[::-1])
    lines = []
    out = PTm.OutputBuffer(w, h, st.device)
    t = PTm.renderLoop(st, out, 4, stats_every=2, stats_out=lines.append)
    assert t["subframes"] == 4 and len(lines) == 2 and st.params.subframe_index == 7 and '"render_ms"' in lines[0]
    PTm.cleanupState(st)


def test_deep_tree_fills_the_traversal_stacks():
    """triangles in geometric progression make the SAH builder peel them off one by one: a tree at the depth cap, rays
    that walk it with full stacks; the entry-pool kernel (stack -> pool conversion near capacity), the streaming kernel
    and the one-thread-per-ray kernel agree with the oracle"""
    torch = _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import scenes
    n = 6000
    rng = np.random.default_rng(21)
    k = np.arange(n)
    scale = (0.5 ** (k / 60.0)).astype(np.float32)                      # 100 octaves of sizes, shrinking towards the origin
    centre = np.stack([scale, 0.3 * scale * np.sin(k * 0.37), 0.3 * scale * np.cos(k * 0.23)], 1).astype(np.float32)
    verts = np.zeros((n * 3, 4), np.float32)
    for c in range(3):
        verts[c::3, :3] = centre + rng.normal(0, 1, (n, 3)).astype(np.float32) * scale[:, None] * 0.2
    scene = {"name": "spiral", "vertices": verts, "mat_index": np.zeros(n, np.uint32), "emission": np.zeros((1, 3), np.float32),
             "diffuse": np.ones((1, 3), np.float32), "bg": np.zeros(4, np.float32),
             "light": scenes.cornell_box()["light"], "camera": {"eye": (2, 0, 0), "lookat": (0, 0, 0), "up": (0, 1, 0), "fovY": 40.0}}
    PT, st = _state(scene, 64, 64)
    info = st.context.accel_info(st.gas_handle)
    assert info["depth"] == 32, info                                      # the builder's depth cap
    osc = _oracle_scene(scene, st)
    n_rays = 60000
    rays = np.zeros((n_rays, 8), np.float32)
    rays[:, :3] = rng.uniform((1.5, -0.3, -0.3), (2.0, 0.3, 0.3), (n_rays, 3))
    zoom = (0.5 ** rng.uniform(0, 30, n_rays))[:, None]                    # aim at every scale of the spiral
    target = np.stack([np.ones(n_rays), rng.uniform(-0.3, 0.3, n_rays), rng.uniform(-0.3, 0.3, n_rays)], 1) * zoom
    rays[:, 4:7] = target - rays[:, :3]
    rays[:, 3], rays[:, 7] = 0.0, 1e16
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.zeros((n_rays, 4), dtype=torch.float32, device="cuda")
    t, prim, u, v = osc.trace_closest(rays, O.MODE_BVH)
    assert (prim != 0xffffffff).mean() > 0.3
    for variant, grid, steal in ((3, 16, 32), (3, 1, 1), (3, 2, 64), (2, 1, 1), (0, 0, 32)):
        st.context.set_option(L.OPT_PERSISTENT, variant)
        st.context.set_option(L.OPT_TRACE_GRID, grid)
        st.context.set_option(L.OPT_STEAL_MIN, steal)
        d_hits.zero_()
        st.context.trace_closest(st.gas_handle, d_rays.data_ptr(), n_rays, d_hits.data_ptr())
        st.context.sync()
        hits = d_hits.cpu().numpy()
        assert np.array_equal(hits[:, 1].view(np.uint32), prim), (variant, grid, steal)
        assert np.array_equal(hits[:, 0].view(np.uint32), t.view(np.uint32)), (variant, grid, steal)
    PT.cleanupState(st)


@pytest.mark.parametrize("name,n,builder", [("cornell", 0, 1), ("soup", 40000, 1), ("soup", 7, 1), ("cornell", 0, 2), ("soup", 40000, 2), ("soup", 130, 2), ("hall", 0, 2),
                                                    ("cornell", 0, 3), ("soup", 40000, 3), ("soup", 130, 3), ("hall", 0, 3),
                                                    ("cornell", 0, 4), ("soup", 40000, 4), ("soup", 130, 4), ("hall", 0, 4), ("dense", 30000, 4)])
def test_device_lbvh_build_gives_the_same_pictures(name, n, builder, tmp_path, monkeypatch):
    """MIPT_OPT_BUILD = 1 / 2: the BVH built on the device (1: Morton codes + radix tree + SAH treelets; 2, round 4: top-down binned SAH
    above the treelets) is a sound BVH over the same triangles (oracle validation), traversal over it agrees with the oracle walking the
    same tree, and frames are bit-identical to those over the host's SAH tree -- hit records do not depend on the tree"""
    torch = _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    monkeypatch.setenv("MIPT_BUILD_HOST_BELOW", "0")          # (the default builder hands inputs below 2048 triangles to the host: here everything is built on the device)
    if name == "hall":      # large triangles among small ones (walls around boxes): where the Morton cut loses most
        scene = scenes.from_gltf(scenes.write_rooms_gltf(tmp_path / "hall.gltf", n_boxes=2000))
    elif name == "dense":   # triangles several cells long: where the builders with spatial splits do cut
        scene = scenes.random_soup(n, edge_scale=0.05)
    else:
        scene = scenes.cornell_box() if name == "cornell" else scenes.random_soup(n)
    w, h = 128, 96
    frames, infos = [], []
    for mode in (0, builder):
        st = PT.PathTracerState()
        PT.createContext(st, 0)
        st.context.set_option(L.OPT_BUILD, mode)
        PT.buildMeshAccel(st, scene)
        PT.createSBT(st)
        PT.initLaunchParams(st, w, h, 2)
        cam = scene["camera"]
        PT.handleCameraUpdate(st, PT.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], 1.0))
        st.context.set_option(L.OPT_MAX_DEPTH, 3)
        acc, frame, stats = _render_gpu(PT, st)
        info = st.context.accel_info(st.gas_handle)
        nodes, tris = st.context.accel_export(st.gas_handle)
        osc = O.Scene(scene).set_bvh(nodes, tris)                       # raises if the tree is not a sound BVH of the scene
        if mode != 0:
            rays = osc.primary_rays(scenes.launch_desc(scene, w, h))[::7]
            d_rays = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
            d_hits = torch.zeros((rays.shape[0], 4), dtype=torch.float32, device="cuda")
            st.context.trace_closest(st.gas_handle, d_rays.data_ptr(), rays.shape[0], d_hits.data_ptr())
            st.context.sync()
            t, prim, u, v = osc.trace_closest(rays, O.MODE_BVH)
            hits = d_hits.cpu().numpy()
            assert np.array_equal(hits[:, 1].view(np.uint32), prim) and np.array_equal(hits[:, 0].view(np.uint32), t.view(np.uint32))
        if mode in (2, 3, 4):
            # the top-down front end moves triangles with atomics (and the host builder with spatial splits runs on several threads); the tree must not depend on where they fell: a second build is the same tree
            PT.buildMeshAccel(st, scene)
            nodes2, tris2 = st.context.accel_export(st.gas_handle)
            assert np.array_equal(nodes.view(np.uint32), nodes2.view(np.uint32)) and np.array_equal(tris.view(np.uint32), tris2.view(np.uint32))
        frames.append((acc, frame, stats)); infos.append(info)
        PT.cleanupState(st)
    assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32))
    assert np.array_equal(frames[0][1], frames[1][1])
    assert frames[0][2]["closest_rays"] == frames[1][2]["closest_rays"] and frames[0][2]["shadow_rays"] == frames[1][2]["shadow_rays"]
    if builder in (3, 4): # spatial splits: triangles referenced by several leaves -- more records than triangles, never fewer
        assert infos[0]["num_triangles"] <= infos[1]["num_triangles"] <= 1.5 * infos[0]["num_triangles"] + 1
        assert infos[1]["depth"] <= 32 and infos[1]["leaf_max"] <= 4
        if name == "dense":
            assert infos[1]["num_triangles"] > 1.05 * infos[0]["num_triangles"], infos
        return
    assert infos[0]["num_triangles"] == infos[1]["num_triangles"] and infos[1]["depth"] <= 32 and infos[1]["leaf_max"] <= 4


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_accel_update_refits_to_moved_vertices(mode, monkeypatch):
    """mipt_accel_update (= OPTIX_BUILD_OPERATION_UPDATE): after the vertices have moved, the refitted tree is a sound BVH of
    the new triangles (oracle validation) and frames over it are bit-identical to frames over a fresh build.  Modes 3 and 4: trees with
    spatial splits (a dense soup, so that triangles are cut) -- a refit gives every record the box of its whole moved triangle"""
    torch = _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    monkeypatch.setenv("MIPT_BUILD_HOST_BELOW", "0")
    scene = scenes.random_soup(20000, edge_scale=0.05 if mode in (3, 4) else 0.01)
    rng = np.random.default_rng(17)
    moved = dict(scene)
    v = np.array(scene["vertices"], np.float32, copy=True)
    v[:, :3] += rng.normal(0.0, 0.004, (v.shape[0], 3)).astype(np.float32)
    tri = np.arange(v.shape[0]) // 3
    v[(tri % 5) == 0, 1] += np.float32(0.05)                      # a fifth of the triangles move as a block
    moved["vertices"] = v
    w, h = 128, 96

    def make(sc):
        st = PT.PathTracerState()
        PT.createContext(st, 0)
        st.context.set_option(L.OPT_BUILD, mode)
        PT.buildMeshAccel(st, sc)
        PT.createSBT(st)
        PT.initLaunchParams(st, w, h, 2)
        cam = sc["camera"]
        PT.handleCameraUpdate(st, PT.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], 1.0))
        st.context.set_option(L.OPT_MAX_DEPTH, 3)
        return st

    st = make(scene)
    before = _render_gpu(PT, st)
    info0 = st.context.accel_info(st.gas_handle)
    assert info0["num_triangles"] > 21000 if mode in (3, 4) else info0["num_triangles"] == 20000
    d_mat = torch.from_numpy(np.ascontiguousarray(scene["mat_index"]).astype(np.int32)).cuda()
    if mode == 4:
        # a refit to the SAME vertices gives the builder's own boxes back: every record's box is clipped to its cell again, plane by plane
        nodes0, tris0 = st.context.accel_export(st.gas_handle)
        st.context.accel_update(st.gas_handle, [{"vertices": st.d_vertices.data_ptr(), "num_vertices": v.shape[0], "stride": 16, "sbt_index_offsets": d_mat.data_ptr()}])
        nodes1, tris1 = st.context.accel_export(st.gas_handle)
        assert np.array_equal(tris0.view(np.uint32), tris1.view(np.uint32))
        diff = (nodes0.view(np.uint32) != nodes1.view(np.uint32)).any(1)
        assert not diff.any(), "%d of %d nodes differ after a refit to the same vertices" % (int(diff.sum()), nodes0.shape[0])
    st.d_vertices.copy_(torch.from_numpy(v))                      # the hit-group records point at this buffer
    st.context.accel_update(st.gas_handle, [{"vertices": st.d_vertices.data_ptr(), "num_vertices": v.shape[0], "stride": 16,
                                             "sbt_index_offsets": d_mat.data_ptr()}])
    refit = _render_gpu(PT, st)
    info1 = st.context.accel_info(st.gas_handle)
    assert (info0["num_nodes"], info0["depth"], info0["num_triangles"]) == (info1["num_nodes"], info1["depth"], info1["num_triangles"])
    nodes, tris = st.context.accel_export(st.gas_handle)
    osc = O.Scene(moved).set_bvh(nodes, tris)                       # raises if the refitted tree is not a sound BVH of the moved scene
    a, f, ost = osc.render(scenes.launch_desc(moved, w, h, 2), max_depth=3, threads=4)
    assert np.array_equal(refit[0].view(np.uint32), a.view(np.uint32)) and np.array_equal(refit[1], f)
    assert refit[2]["closest_rays"] == ost["closest_rays"] and refit[2]["shadow_rays"] == ost["shadow_rays"]
    assert not np.array_equal(before[0], refit[0])
    with pytest.raises(Exception):
        st.context.accel_update(st.gas_handle, [{"vertices": st.d_vertices.data_ptr(), "num_vertices": v.shape[0] - 3, "stride": 16}])
    with pytest.raises(Exception):
        st.context.accel_update(st.gas_handle + 99, [{"vertices": st.d_vertices.data_ptr(), "num_vertices": v.shape[0], "stride": 16}])
    PT.cleanupState(st)
    st2 = make(moved)
    fresh = _render_gpu(PT, st2)
    PT.cleanupState(st2)
    assert np.array_equal(refit[0].view(np.uint32), fresh[0].view(np.uint32)) and np.array_equal(refit[1], fresh[1])
    assert refit[2]["closest_rays"] == fresh[2]["closest_rays"] and refit[2]["shadow_rays"] == fresh[2]["shadow_rays"]


@pytest.mark.parametrize("builder", [1, 4])
def test_entry_pool_beyond_four_million_triangles(builder):
    """the pool's packed entries address 2^23 triangles: a scene past the former 2^22 limit still runs on the default
    kernels and agrees with the oracle walking the same tree.  builder 4: spatial splits add records up to that limit and no further
    (the reference budget is clipped to the index width)"""
    _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    n = (1 << 22) + 150_000 if builder == 1 else 6_000_000
    scene = scenes.random_soup(n, edge_scale=0.004 if builder == 1 else 0.008)
    st = PT.PathTracerState()
    PT.createContext(st, 0)
    st.context.set_option(L.OPT_BUILD, builder)               # the device builders: milliseconds
    PT.buildMeshAccel(st, scene)
    PT.createSBT(st)
    w, h = 96, 64
    PT.initLaunchParams(st, w, h, 1)
    cam = scene["camera"]
    PT.handleCameraUpdate(st, PT.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], 1.0))
    st.context.set_option(L.OPT_MAX_DEPTH, 2)
    info = st.context.accel_info(st.gas_handle)
    if builder == 4:           # 6 M triangles ask for 3 M more references; 2^23 - 1 - 6 M are granted
        assert n + 1_000_000 < info["num_triangles"] < (1 << 23) and info["leaf_max"] <= 4, info
    else:
        assert info["num_triangles"] == n and info["leaf_max"] <= 4
    acc, frame, stats = _render_gpu(PT, st)
    nodes, tris = st.context.accel_export(st.gas_handle)
    assert int(tris[:, 3].view(np.uint32).max()) == n - 1
    osc = O.Scene(scene).set_bvh(nodes, tris)
    a, f, ost = osc.render(scenes.launch_desc(scene, w, h, 1), max_depth=2, threads=8)
    _assert_pixels(acc, a)
    assert np.array_equal(frame, f)
    assert stats["closest_rays"] == ost["closest_rays"] and stats["shadow_rays"] == ost["shadow_rays"]
    st.context.set_option(L.OPT_PERSISTENT, 2)                # and the streaming variant on the same tree
    acc2, frame2, _ = _render_gpu(PT, st)
    assert np.array_equal(acc2.view(np.uint32), acc.view(np.uint32))
    PT.cleanupState(st)


@pytest.mark.parametrize("kind", ["none", "one", "two", "five_with_degenerates"])
def test_tiny_and_degenerate_scenes(kind):
    """0, 1, 2 triangles (the forced root with an empty leaf), zero-area and duplicated triangles: the launch terminates and
    equals the oracle"""
    _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    base = scenes.cornell_box()
    quad = np.asarray([[-200, 100, 300], [700, 100, 300], [250, 500, 300],          # a big triangle in front of the camera
                       [0, 0, 100], [556, 0, 100], [278, 548, 500]], np.float32)
    tris = {"none": np.zeros((0, 3), np.float32), "one": quad[:3], "two": quad,
            "five_with_degenerates": np.concatenate([quad, quad[:3],                            # a duplicate
                                                     np.asarray([[100, 100, 200]] * 3, np.float32),       # a point
                                                     np.asarray([[0, 0, 50], [100, 100, 50], [200, 200, 50]], np.float32)])}[kind]  # a line
    scene = dict(base)
    v = np.zeros((tris.shape[0], 4), np.float32); v[:, :3] = tris
    scene["vertices"] = v
    scene["mat_index"] = (np.arange(tris.shape[0] // 3) % 4).astype(np.uint32)
    w, h = 64, 48
    if kind == "none":
        # an empty vertex buffer is a null device pointer: the hit-group records are refused, not dereferenced
        with pytest.raises(Exception, match="null vertex pointer"):
            _state(scene, w, h, 2, 4)
        return
    PT2, st = _state(scene, w, h, 2, 4)
    info = st.context.accel_info(st.gas_handle)
    assert info["num_triangles"] == tris.shape[0] // 3
    acc, frame, stats = _render_gpu(PT2, st)
    osc = _oracle_scene(scene, st)
    a, f, ost = osc.render(scenes.launch_desc(scene, w, h, 2), max_depth=4, threads=2)
    _assert_pixels(acc, a)
    assert np.array_equal(frame, f)
    assert stats["closest_rays"] == ost["closest_rays"] and stats["shadow_rays"] == ost["shadow_rays"]
    for variant in (0, 2):
        st.context.set_option(L.OPT_PERSISTENT, variant)
        acc2, _, _ = _render_gpu(PT2, st)
        assert np.array_equal(acc2.view(np.uint32), acc.view(np.uint32)), variant
    PT2.cleanupState(st)


def test_non_finite_rays_terminate_and_agree():
    """NaN / infinite / zero ray components: every traversal kernel terminates and all of them -- and the oracle -- report
    the same hits (comparisons with NaN are false everywhere, min / max return the non-NaN operand on both sides)"""
    torch = _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import scenes
    scene = scenes.random_soup(30000)
    PT, st = _state(scene, 32, 32)
    osc = _oracle_scene(scene, st)
    rng = np.random.default_rng(23)
    n = 4096
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = rng.uniform(-0.2, 1.2, (n, 3))
    rays[:, 4:7] = rng.normal(size=(n, 3))
    rays[:, 3], rays[:, 7] = 0.01, 1e16
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    k = 0
    for col, val in ((0, nan), (4, nan), (5, inf), (6, -inf), (3, nan), (7, nan), (7, inf), (3, -inf), (0, inf), (1, -inf)):
        rays[k:k + 64, col] = val; k += 64
    rays[k:k + 64, 4:7] = 0.0; k += 64                                   # zero direction
    rays[k:k + 64, 4:7] = [inf, inf, inf]; k += 64
    rays[k:k + 64, 4:7] = nan; k += 64
    rays[k:k + 64, 3] = -1.0; k += 64                                    # negative tmin
    d_rays = torch.from_numpy(rays).cuda()
    results = {}
    for variant in (3, 2, 0):
        st.context.set_option(L.OPT_PERSISTENT, variant)
        d_hits = torch.full((n, 4), 7.0, dtype=torch.float32, device="cuda")
        d_occ = torch.full((n,), 9, dtype=torch.int32, device="cuda")
        st.context.trace_closest(st.gas_handle, d_rays.data_ptr(), n, d_hits.data_ptr())
        st.context.trace_any(st.gas_handle, d_rays.data_ptr(), n, d_occ.data_ptr())
        st.context.sync()
        results[variant] = (d_hits.cpu().numpy(), d_occ.cpu().numpy())
    h3, o3 = results[3]
    for variant in (2, 0):
        hv, ov = results[variant]
        assert np.array_equal(hv[:, :2].view(np.uint32), h3[:, :2].view(np.uint32)), variant
        assert np.array_equal(ov, o3), variant
    t, prim, u, v = osc.trace_closest(rays, O.MODE_BVH)
    assert np.array_equal(h3[:, 1].view(np.uint32), prim)
    assert np.array_equal(h3[:, 0].view(np.uint32), t.view(np.uint32))
    occ = osc.trace_any(rays, O.MODE_BVH)
    assert np.array_equal(o3 != 0, np.asarray(occ) != 0)
    assert (prim[k:] != 0xffffffff).mean() > 0.05                        # the ordinary rays still hit things
    PT.cleanupState(st)


def test_device_build_reads_every_input_form():
    """indexed u16 / u32 / non-indexed inputs, strided vertices, a 3x4 transform, SBT offsets + geometry flags, host and
    device pointers: the device-side input assembly makes the same triangles as the host builder's"""
    torch = _torch()
    from optix_raytracer_amd import _lib as L, api
    rng = np.random.default_rng(3)
    nv = 300
    verts = np.zeros((nv, 6), np.float32)                       # stride 24: positions + junk
    verts[:, :3] = rng.uniform(-1, 1, (nv, 3))
    idx16 = rng.integers(0, nv, (150, 3)).astype(np.uint16)
    idx32 = rng.integers(0, nv, (90, 3)).astype(np.uint32)
    soup = rng.uniform(-2, 2, (60 * 3, 3)).astype(np.float32)
    xf = np.asarray([[0.5, 0.1, 0, 1.0], [-0.1, 0.7, 0.2, -2.0], [0.0, 0.3, 0.9, 0.5]], np.float32)
    sbt = rng.integers(0, 3, 150).astype(np.uint32)
    flags = np.asarray([1, 0, 6], np.uint32)
    d_verts = torch.from_numpy(verts).cuda()
    d_idx32 = torch.from_numpy(idx32.astype(np.int32)).cuda()
    inputs = [
        {"vertices": d_verts.data_ptr(), "num_vertices": nv, "stride": 24, "indices": idx16, "index_format": 2, "num_triangles": 150,
         "sbt_index_offsets": sbt, "flags": flags},
        {"vertices": verts, "num_vertices": nv, "stride": 24, "indices": d_idx32.data_ptr(), "index_format": 4, "num_triangles": 90, "transform": xf},
        {"vertices": soup, "num_vertices": 180, "stride": 12},
    ]
    out = []
    for mode in (0, 1):
        ctx = api.Context(0)
        ctx.set_option(L.OPT_BUILD, mode)
        h = ctx.accel_build(inputs)
        info = ctx.accel_info(h)
        assert info["num_triangles"] == 300
        nodes, tris = ctx.accel_export(h)
        t = np.frombuffer(tris.tobytes(), np.uint32).reshape(-1, 12)
        out.append(t[np.argsort(t[:, 3])])                       # by original primitive index
        ctx.destroy()
    assert np.array_equal(out[0], out[1])
    assert set(out[1][:150, 7]) <= {1, 0, 6} and (out[1][150:, 7] == 0).all()
    # an index past the vertex array is an error on both paths
    bad = dict(inputs[0], indices=np.full((150, 3), 999, np.uint16))
    for mode in (0, 1):
        ctx = api.Context(0)
        ctx.set_option(L.OPT_BUILD, mode)
        with pytest.raises(api.MiptError, match="out of range"):
            ctx.accel_build([bad])
        ctx.destroy()


def test_packed_fp16_nodes_give_the_same_hits(monkeypatch):
    """MIPT_OPT_NODE_FORMAT: format 1 rounds every child box outward to fp16 values and gives the entry-pool kernels a packed
    32-byte node (two 16-byte loads per visit instead of four); format 2 (the default) gives them a BVH4 collapse of that tree
    (four fp16 boxes per 64-byte node, about 0.6 x the steps per ray).  Wider boxes only weaken pruning: frames, ray counters and hit
    records equal those over the fp32 boxes bit for bit; the exported nodes ARE the fp16 values (the oracle walks the same
    boxes, so visit counters agree too); a scene that does not suit fp16 boxes keeps format 0."""
    torch = _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import scenes
    scene = scenes.random_soup(60000)
    w, h = 320, 180
    launch = scenes.launch_desc(scene, w, h, 2)
    got = {}
    for fmt in (0, 1, 2):
        monkeypatch.setenv("MIPT_NODE_FORMAT", str(fmt))
        PT, st = _state(scene, w, h, spp=2)
        info = st.context.accel_info(st.gas_handle)
        assert info["node_format"] == fmt
        nodes, tris = st.context.accel_export(st.gas_handle)
        acc, frame, stats = _render_gpu(PT, st)
        osc = O.Scene(scene).set_bvh(nodes, tris)
        a, f, ost = osc.render(launch, threads=4)
        _assert_pixels(acc, a)
        assert np.array_equal(frame, f)
        assert stats["closest_rays"] == ost["closest_rays"] and stats["shadow_rays"] == ost["shadow_rays"]
        # counting launch: the per-ray order of the oracle over the SAME boxes
        st.context.set_option(L.OPT_COUNT_VISITS, 1)
        st.context.set_option(L.OPT_MAX_DEPTH, 2)
        acc2, _, stats2 = _render_gpu(PT, st)
        a2, _, ost2 = osc.render(launch, max_depth=2, threads=4)
        _assert_pixels(acc2, a2)
        assert stats2["node_visits"] == ost2["node_visits"] and stats2["tri_tests"] == ost2["tri_tests"]
        # ray-buffer queries through the pool kernels
        rays = osc.primary_rays(launch)[:: 3]
        d_rays = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
        d_hits = torch.zeros((rays.shape[0], 4), dtype=torch.float32, device="cuda")
        st.context.set_option(L.OPT_COUNT_VISITS, 0)
        st.context.trace_closest(st.gas_handle, d_rays.data_ptr(), rays.shape[0], d_hits.data_ptr())
        st.context.sync()
        got[fmt] = (acc, nodes, d_hits.cpu().numpy(), stats, stats2)
        PT.cleanupState(st)
    for fmt in (1, 2):
        assert np.array_equal(got[0][0].view(np.uint32), got[fmt][0].view(np.uint32)), fmt
        assert np.array_equal(got[0][2].view(np.uint32), got[fmt][2].view(np.uint32)), fmt
        assert got[0][3]["closest_rays"] == got[fmt][3]["closest_rays"] and got[0][3]["shadow_rays"] == got[fmt][3]["shadow_rays"]
    # format 2 (the BVH4 mirror) is a collapse of format 1's tree: the exported BVH2 is the same, node for node
    assert np.array_equal(got[1][1].view(np.uint32), got[2][1].view(np.uint32))
    assert got[1][4] == got[2][4]
    n0, n1 = got[0][1], got[1][1]
    assert np.array_equal(n0[:, 12:].view(np.uint32), n1[:, 12:].view(np.uint32))          # same tree
    box1 = n1[:, :12]
    assert np.array_equal(box1.astype(np.float16).astype(np.float32), box1)                # every plane is an fp16 value
    lo, hi = [0, 2, 4, 6, 8, 10], [1, 3, 5, 7, 9, 11]
    assert (n1[:, lo] <= n0[:, lo]).all() and (n1[:, hi] >= n0[:, hi]).all()               # rounded outward
    assert (n0[:, lo] - n1[:, lo]).max() <= 2.0 ** -10 and (n1[:, hi] - n0[:, hi]).max() <= 2.0 ** -10    # by less than one fp16 step at |x| <= 2
    assert got[1][4]["node_visits"] >= got[0][4]["node_visits"]
    assert got[1][4]["node_visits"] <= 1.1 * got[0][4]["node_visits"]
    # a scene 5000 units from the origin, 1 unit across: fp16 steps of 4 units would make the boxes useless -> fp32 kept
    monkeypatch.setenv("MIPT_NODE_FORMAT", "2")
    far = dict(scene)
    v = np.array(scene["vertices"], np.float32, copy=True)
    v[:, :3] += np.float32(5000.0)
    far["vertices"] = v
    cam = dict(scene["camera"])
    cam["eye"] = tuple(np.float32(x) + np.float32(5000.0) for x in cam["eye"])
    cam["lookat"] = tuple(np.float32(x) + np.float32(5000.0) for x in cam["lookat"])
    far["camera"] = cam
    lt = dict(scene["light"])
    lt["corner"] = np.asarray(lt["corner"], np.float32) + np.float32(5000.0)
    far["light"] = lt
    PT, st = _state(far, 160, 96)
    assert st.context.accel_info(st.gas_handle)["node_format"] == 0
    acc, frame, stats = _render_gpu(PT, st)
    a, f, ost = _oracle_scene(far, st).render(scenes.launch_desc(far, 160, 96), threads=4)
    _assert_pixels(acc, a)
    PT.cleanupState(st)


def test_in_process_group_renders_the_full_frame():
    """mipt_group_create / launch / allgather / sync (the one-host-thread form of SDK/optixMultiGPU, optixMultiGPU.cpp:562-594)
    on the devices this box has: the member's shard goes through the group's exchange (RCCL's ncclAllGather when librccl loads,
    also with ONE member) and the scatter kernel, and the assembled frame equals the single-launch frame and the oracle's."""
    torch = _torch()
    from optix_raytracer_amd import api
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    n_dev = min(torch.cuda.device_count(), 2)
    scene = scenes.cornell_box()
    w, h, spp = 200, 120, 2
    grp = api.Group(list(range(n_dev)))
    assert len(grp.contexts) == n_dev
    states = []
    for i, ctx in enumerate(grp.contexts):              # the sample replicates the scene per device (optixMultiGPU.cpp:1093-1102)
        st = PT.PathTracerState()
        st.device = torch.device("cuda", grp.devices[i])
        torch.cuda.set_device(st.device)
        st.context = ctx
        PT.buildMeshAccel(st, scene)
        PT.createSBT(st)
        PT.initLaunchParams(st, w, h, spp)
        cam = scene["camera"]
        PT.handleCameraUpdate(st, PT.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], 1.0))
        st.frame = torch.zeros((h, w, 4), dtype=torch.uint8, device=st.device)
        states.append(st)
    for sub in range(2):                                # progressive: the compact shard buffers carry the running average
        for st in states:
            st.params.subframe_index = sub
        grp.launch([st.params for st in states], w, h)
        grp.allgather([st.accum.data_ptr() for st in states], [st.frame.data_ptr() for st in states])
        grp.sync()
    print("group exchange through RCCL:", grp.uses_rccl)
    # the same two subframes by plain launches, and by the oracle
    PT2, ref = _state(scene, w, h, spp)
    for sub in range(2):
        acc_ref, frame_ref, _ = _render_gpu(PT2, ref, subframe=sub)
    osc = _oracle_scene(scene, ref)
    a = None
    for sub in range(2):
        a, f, _ = osc.render(scenes.launch_desc(scene, w, h, spp, subframe=sub), accum=a)
    for st in states:
        got = st.accum.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), acc_ref.view(np.uint32))
        assert np.array_equal(got.view(np.uint32), a.view(np.uint32))
        assert np.array_equal(st.frame.cpu().numpy(), frame_ref)
    with pytest.raises(api.MiptError):
        grp.launch([states[0].params] * n_dev, w + 8, h)             # size does not match the params
    PT2.cleanupState(ref)
    grp.destroy()


def _soup_scene(verts):
    from optix_raytracer_amd import scenes
    base = scenes.random_soup(8)
    n = verts.shape[0] // 3
    sc = dict(base)
    sc["vertices"] = np.ascontiguousarray(verts, np.float32)
    sc["mat_index"] = np.zeros(n, np.uint32)
    sc["name"] = "degenerate%d" % n
    return sc


@pytest.mark.parametrize("builder", [2, 4])
@pytest.mark.parametrize("kind", ["identical", "line", "two_clusters", "spikes", "big_and_small"])
def test_device_sah_builder_on_degenerate_inputs(kind, builder, monkeypatch):
    """The device SAH builder (the default) where binning has nothing to hold on to: thousands of copies of ONE triangle (no centroid
    extent on any axis: median splits all the way down), triangles along a line (one usable axis), two far clusters (30 of 32 bins
    empty at the root), geometrically shrinking spikes (the SAH peels them off one by one: deeper than the stack, so the build falls back
    to the host builder, which caps the depth), large triangles among small ones.  Each tree must be a sound BVH of its scene (oracle
    validation), and hit records over it equal the brute-force intersector's."""
    torch = _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT
    rng = np.random.default_rng(21)
    if kind == "identical":
        tri = np.array([[0.2, 0.2, 0.5, 0], [0.8, 0.2, 0.5, 0], [0.5, 0.8, 0.5, 0]], np.float32)
        verts = np.tile(tri, (3000, 1))
    elif kind == "line":
        n = 4000
        c = np.zeros((n, 1, 3), np.float32); c[:, 0, 0] = np.linspace(0, 1, n)
        c[:, 0, 1] = 0.5; c[:, 0, 2] = 0.5
        off = np.array([[0, -0.01, 0], [0, 0.01, 0], [0, 0, 0.02]], np.float32)      # every centroid on the line y = 0.5, z = 0.5 + 0.02 / 3
        verts = np.concatenate([(c + off).reshape(-1, 3), np.zeros((3 * n, 1), np.float32)], 1)
    elif kind == "two_clusters":
        n = 6000
        c = rng.uniform(0, 0.01, (n, 1, 3)).astype(np.float32); c[n // 2:, 0, 0] += 0.99
        verts = np.concatenate([(c + rng.uniform(-0.002, 0.002, (n, 3, 3))).reshape(-1, 3), np.zeros((3 * n, 1))], 1).astype(np.float32)
    elif kind == "spikes":
        n = 400
        tri = np.zeros((n, 3, 4), np.float32)
        for i in range(n):
            s = 0.9 ** i
            tri[i, :, :3] = np.array([1.0 - s, 0.5, 0.5]) + s * 0.4 * rng.normal(size=(3, 3))
        verts = tri.reshape(-1, 4)
    else:
        n = 20000
        c = rng.uniform(0, 1, (n, 1, 3))
        size = np.where(rng.uniform(size=(n, 1, 1)) < 0.01, 0.5, 0.004)
        verts = np.concatenate([(c + size * rng.uniform(-1, 1, (n, 3, 3))).reshape(-1, 3), np.zeros((3 * n, 1))], 1).astype(np.float32)
    scene = _soup_scene(verts)
    monkeypatch.setenv("MIPT_BUILD_HOST_BELOW", "0")
    st = PT.PathTracerState()
    PT.createContext(st, 0)
    assert st.context.get_option(L.OPT_BUILD) == 2
    st.context.set_option(L.OPT_BUILD, builder)                        # (4: with spatial splits -- cuts through coincident, collinear, far-apart and oversized triangles)
    PT.buildMeshAccel(st, scene)
    info = st.context.accel_info(st.gas_handle)
    n_tris = verts.shape[0] // 3
    assert (info["num_triangles"] == n_tris if builder == 2 else n_tris <= info["num_triangles"] <= n_tris + n_tris // 2) and info["depth"] <= 32 and info["leaf_max"] <= 4
    nodes, tris = st.context.accel_export(st.gas_handle)
    osc = O.Scene(scene).set_bvh(nodes, tris)                          # raises if the tree is not a sound BVH of the scene
    n_rays = 20000
    rays = np.zeros((n_rays, 8), np.float32)
    rays[:, :3] = rng.uniform(-0.5, 1.5, (n_rays, 3)); rays[:, 2] = -1.0
    cen = verts[:, :3].reshape(-1, 3, 3).mean(1)                       # half the rays aim at a triangle, half anywhere in the unit cube
    tgt = np.where(rng.uniform(size=(n_rays, 1)) < 0.5, cen[rng.integers(0, cen.shape[0], n_rays)] + rng.normal(0, 1e-3, (n_rays, 3)), rng.uniform(0, 1, (n_rays, 3)))
    d = tgt - rays[:, :3]
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 3], rays[:, 7] = 1e-3, 1e16
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.zeros((n_rays, 4), dtype=torch.float32, device="cuda")
    st.context.trace_closest(st.gas_handle, d_rays.data_ptr(), n_rays, d_hits.data_ptr())
    st.context.sync()
    hits = d_hits.cpu().numpy()
    t, prim, u, v = osc.trace_closest(rays, O.MODE_BRUTE)
    assert np.array_equal(hits[:, 1].view(np.uint32), prim) and np.array_equal(hits[:, 0].view(np.uint32), t.view(np.uint32))
    assert (prim != 0xffffffff).sum() > 50
    PT.cleanupState(st)


@pytest.mark.parametrize("name", ["soup", "hall"])
def test_forced_spatial_splits_lose_no_hit(name, tmp_path, monkeypatch):
    """MIPT_SBVH_FORCE=1 with a reference budget of four times the triangle count: the host builder takes EVERY valid spatial split, so
    the tree is full of triangles referenced from several leaves through small clipped boxes -- the hardest case for the claim that the
    clipped bounds are conservative.  Frames, counters and 60 k hit records must equal those over the plain host tree bit for bit, the
    records also the brute-force intersector's."""
    torch = _torch()
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import pathtracer as PT
    from optix_raytracer_amd import scenes
    # (a soup of triangles five times as large as the gaps between them; the hall: walls and a floor around 1 500 boxes)
    scene = scenes.random_soup(30000, edge_scale=0.05) if name == "soup" else scenes.from_gltf(scenes.write_rooms_gltf(tmp_path / "hall.gltf", n_boxes=1500))
    w, h = 160, 120
    out = {}
    for mode in (0, 3):
        if mode == 3:
            monkeypatch.setenv("MIPT_SBVH_FORCE", "1"); monkeypatch.setenv("MIPT_SBVH_EXTRA", "4")
        st = PT.PathTracerState()
        PT.createContext(st, 0)
        st.context.set_option(L.OPT_BUILD, mode)
        PT.buildMeshAccel(st, scene)
        PT.createSBT(st)
        PT.initLaunchParams(st, w, h, 2)
        cam = scene["camera"]
        PT.handleCameraUpdate(st, PT.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], 1.0))
        st.context.set_option(L.OPT_MAX_DEPTH, 4)
        acc, frame, stats = _render_gpu(PT, st)
        info = st.context.accel_info(st.gas_handle)
        nodes, tris = st.context.accel_export(st.gas_handle)
        osc = O.Scene(scene).set_bvh(nodes, tris)
        rays = osc.primary_rays(scenes.launch_desc(scene, 320, 200))[::1]
        rng = np.random.default_rng(3)
        extra = np.zeros((20000, 8), np.float32)
        lo, hi = scene["vertices"][:, :3].min(0), scene["vertices"][:, :3].max(0)
        extra[:, :3] = lo + rng.uniform(0, 1, (20000, 3)) * (hi - lo)
        extra[:, 4:7] = rng.normal(size=(20000, 3)); extra[:, 4:7] /= np.linalg.norm(extra[:, 4:7], axis=1, keepdims=True)
        extra[:, 3], extra[:, 7] = 1e-4, 1e16
        rays = np.ascontiguousarray(np.concatenate([rays[:40000], extra]), np.float32)
        d_rays = torch.from_numpy(rays).cuda()
        d_hits = torch.zeros((rays.shape[0], 4), dtype=torch.float32, device="cuda")
        st.context.trace_closest(st.gas_handle, d_rays.data_ptr(), rays.shape[0], d_hits.data_ptr()); st.context.sync()
        out[mode] = (acc, frame, stats, d_hits.cpu().numpy(), info)
        if mode == 3:
            assert info["num_triangles"] > 1.15 * out[0][4]["num_triangles"]             # shared references all over the tree
            t, prim, u, v = osc.trace_closest(rays, O.MODE_BRUTE)
            hg = out[3][3]
            assert np.array_equal(hg[:, 1].view(np.uint32), prim) and np.array_equal(hg[:, 0].view(np.uint32), t.view(np.uint32))
        PT.cleanupState(st)
    assert np.array_equal(out[0][0].view(np.uint32), out[3][0].view(np.uint32)) and np.array_equal(out[0][1], out[3][1])
    assert out[0][2]["closest_rays"] == out[3][2]["closest_rays"] and out[0][2]["shadow_rays"] == out[3][2]["shadow_rays"]
    assert np.array_equal(out[0][3].view(np.uint32), out[3][3].view(np.uint32))
