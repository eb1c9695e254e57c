#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's headline configuration.

Workload (configs[3], SURVEY.md 8d "C4"): 1,000,000 random triangles in the unit cube, 1920x1080,
spp = 1, subframe 0, max depth 1 (the primary ray and its next-event-estimation shadow ray).
A "step" is one frame: raygen -> trace_closest -> shade -> trace_any -> resolve.  With N > 1 ranks
(one process per GPU, launched by torch.distributed.run) every rank renders its interleaved 8x4-tile
shard of the SAME frame, the compact float4 shards are all-gathered over RCCL and un-permuted
(strong scaling: the total work is the frame).

Prints ONE JSON line on rank 0:  value = (closest + shadow rays actually traced by all ranks) / time.
Scene generation and the BVH build are outside the timed region; inputs are resident
in HBM when the timed region starts.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP, N_TRIS, MAX_DEPTH = 1920, 1080, 1, 1_000_000, 1
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tris", type=int, default=N_TRIS, help="debug only: any other value is not the headline workload")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--max-depth", type=int, default=MAX_DEPTH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", type=str, default="nccl", help="debug: 'gloo' lets several ranks share one GPU (functional check only)")
    ap.add_argument("--persistent", type=int, default=-1, help="tuning: traversal kernel variant (3 entry pool, 2 streaming waves, 0 one thread per ray)")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="N > 1: all-gather + scatter on the launch stream instead of overlapping the next frame's tracing")
    ap.add_argument("--verify", action="store_true", help="debug: compare the gathered frame with a single-launch frame, bit for bit")
    ap.add_argument("--force-shard-path", action="store_true", help="debug: run the N > 1 code path (shard, gather, scatter) on one GPU")
    ap.add_argument("--grid", type=int, default=0, help="tuning: persistent workgroups per CU")
    ap.add_argument("--refill-min", type=int, default=0)
    ap.add_argument("--steal-min", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the bounded CPU-baseline sample")
    return ap.parse_args()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    written by tools/prof_traffic.sh + tools/traffic_json.py): 2*FETCH_SIZE + WRITE_SIZE, in bytes -- FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950.  None when no profile of this kernel is on file."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        e = d["kernels"][kernel]
        return int((2.0 * e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024)
    except Exception:
        return None


def effective_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box
    shows every host thread in the mask but grants a share of them)."""
    n = max(1, len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(st, scene, width, height, max_depth, budget_s):
    """The CPU oracle (oracle/pt_oracle.c, kind "port": this build's C restatement, one single-threaded
    traversal per core) rendering the same frame over the same BVH, repeated until ~budget_s of wall
    time has passed (bounded sample; the BVH export/validation is outside the timing)."""
    import ctypes as C
    import threading

    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from optix_raytracer_amd import scenes
    nodes, tris = st.context.accel_export(st.gas_handle)
    osc = O.Scene(scene).set_bvh(nodes, tris)
    launch = scenes.launch_desc(scene, width, height, SPP)
    cores = effective_cpus()
    accum = np.zeros((height, width, 4), np.float32)      # must outlive the render calls
    p = osc.make_params(launch, accum, None)
    lib = O.lib()
    n_bands = (height + 3) // 4
    stats = [O.OrcStats() for _ in range(cores)]
    lock = threading.Lock()
    state = {"next": 0, "stop": False}
    t_start = time.perf_counter()

    def work(tid):
        while True:
            with lock:                                  # work queue of 4-row bands, frame after frame
                if state["stop"]:
                    return
                b = state["next"]
                state["next"] += 1
                if b % n_bands == 0 and b > 0 and time.perf_counter() - t_start > budget_s:
                    state["stop"] = True
                    state["next"] = b
                    return
            y = (b % n_bands) * 4
            lib.orc_render_rows(osc._h, O.MODE_BVH, C.byref(p), max_depth, y, y + 4, C.byref(stats[tid]))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.perf_counter() - t_start
    rays = sum(s.closest_rays + s.shadow_rays for s in stats)
    frames = state["next"] / n_bands
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": "%.2f frames of the same %dx%d workload (%d rays) in %.1f s wall, %d threads pulling 4-row bands" % (
                frames, width, height, rays, dt, cores)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import api, scenes
    from optix_raytracer_amd import dist as D
    from optix_raytracer_amd import pathtracer as PT

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()      # debug: ranks may share a device
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL
        else:
            dist.init_process_group(args.backend)

    w, h = args.width, args.height
    scene = scenes.random_soup(args.tris)
    t0 = time.perf_counter()
    st = PT.make_state(scene, w, h, SPP, local_rank, max_depth=args.max_depth)
    build_s = time.perf_counter() - t0
    info = st.context.accel_info(st.gas_handle)
    if args.persistent >= 0:
        st.context.set_option(L.OPT_PERSISTENT, args.persistent)
    if args.grid:
        st.context.set_option(L.OPT_TRACE_GRID, args.grid)
    if args.refill_min:
        st.context.set_option(L.OPT_REFILL_MIN, args.refill_min)
    if args.steal_min:
        st.context.set_option(L.OPT_STEAL_MIN, args.steal_min)
    stream = torch.cuda.current_stream()
    frame = torch.zeros((h, w, 4), dtype=torch.uint8, device=st.device)
    st.params.frame_buffer = frame.data_ptr()
    st.params.subframe_index = 0
    sharded = world > 1 or args.force_shard_path
    if sharded:
        # two compact / gathered buffers: the all-gather + scatter of frame k run on a second stream while the
        # main stream already traces frame k+1 (a progressive renderer's steady state); every step's gather
        # and scatter still complete inside the timed region (fence = barrier + device synchronize)
        ns = api.shard_num_samples(w, h, world)
        compact = [torch.zeros((ns, 4), dtype=torch.float32, device=st.device) for _ in range(2)]
        gathered = [torch.zeros((world * ns, 4), dtype=torch.float32, device=st.device) for _ in range(2)]
        xstream = torch.cuda.Stream(device=st.device) if args.overlap else stream
        rendered = [torch.cuda.Event() for _ in range(2)]
        consumed = [torch.cuda.Event() for _ in range(2)]
    state = {"k": 0}

    def step():
        if not sharded:
            st.context.launch(st.params, w, h, 1, stream=stream.cuda_stream)
            return
        b = state["k"] & 1
        state["k"] += 1
        if state["k"] > 2:
            stream.wait_event(consumed[b])                      # frame k-2's gather has read compact[b]
        st.context.launch_shard(st.params, w, h, rank, world, compact[b].data_ptr(), stream=stream.cuda_stream)
        rendered[b].record(stream)
        with torch.cuda.stream(xstream):
            xstream.wait_event(rendered[b])
            if world == 1:
                gathered[b].copy_(compact[b])
            elif args.backend == "nccl":
                dist.all_gather_into_tensor(gathered[b], compact[b])      # the path's one exchange (RCCL over xGMI)
            else:
                gathered[b].copy_(D.all_gather_shards(compact[b].cpu(), world))   # debug backend: staged through the host
            consumed[b].record(xstream)
            st.context.unpermute(gathered[b].data_ptr(), world, w, h, st.accum.data_ptr(), frame.data_ptr(), stream=xstream.cuda_stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- one counting launch (outside the timed region): rays, node visits, triangle tests per frame
    st.context.set_option(L.OPT_COUNT_VISITS, 1)
    step()
    torch.cuda.synchronize()
    counts = st.context.launch_stats()
    st.context.set_option(L.OPT_COUNT_VISITS, 0)

    for _ in range(args.warmup):
        step()
    # ---- timed region: exactly K steps between barrier + synchronize.  HIP events bracket the two traversal
    # kernels on the launch stream (4 records per frame; bracketing all kernels costs 3 % of the frame)
    st.context.set_option(L.OPT_PROFILE, 2)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    tm = st.context.launch_timings()              # event sums over all K timed steps (read after the fence)
    k_closest, k_any = (tm[k] / args.steps for k in ("ms_trace_closest", "ms_trace_any"))
    # ---- per-kernel breakdown of the other kernels: a few more steps with every kernel bracketed, untimed
    st.context.set_option(L.OPT_PROFILE, 1)
    nb = min(args.steps, 5)
    for _ in range(nb):
        step()
    fence()
    tb = st.context.launch_timings()
    k_shade = tb["ms_shade"] / nb
    k_other = (tb["ms_raygen"] + tb["ms_resolve"]) / nb
    st.context.set_option(L.OPT_PROFILE, 0)

    if args.verify and sharded:
        got = st.accum.clone()
        st.context.launch(st.params, w, h, 1, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        same = bool(torch.equal(got.view(torch.int32), st.accum.view(torch.int32)))
        print("verify: sharded + gathered frame %s the single-launch frame" % ("==" if same else "!="), file=sys.stderr)
        if not same:
            raise SystemExit("bench.py --verify: frames differ")

    rays_local = counts["closest_rays"] + counts["shadow_rays"]
    t_max = elapsed
    rays_total = rays_local
    if world > 1:
        red_dev = st.device if args.backend == "nccl" else "cpu"
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
        rr = torch.tensor([rays_local], dtype=torch.int64, device=red_dev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays_total = int(rr.item())

    if rank == 0:
        ms_per_step = t_max / args.steps * 1e3
        value = rays_total * args.steps / t_max / 1e6
        # ---- roofline of the dominant kernel (algorithmic bytes: SURVEY.md 8d / DESIGN.md)
        nv_c = counts["node_visits"] - counts["shadow_node_visits"]
        tt_c = counts["tri_tests"] - counts["shadow_tri_tests"]
        bytes_closest = 64 * nv_c + 48 * tt_c + (32 + 16) * counts["closest_rays"]
        bytes_any = 64 * counts["shadow_node_visits"] + 48 * counts["shadow_tri_tests"] + (32 + 16 + 32) * counts["shadow_rays"]
        variant = {3: "k_trace_pool<%%d, %d>" % (24 if info["depth"] <= 24 else 32), 2: "k_trace_stream<%d, false>",
                   0: "k_trace_simple<%d, false>"}[int(st.context.get_option(L.OPT_PERSISTENT))]
        # a pass of at most MIPT_OPT_TAIL_PATHS paths (a rank's shard at N >= 2) runs in the path kernel: closest-hit and
        # any-hit traversal, shading and accumulation of a wave's 64 paths in ONE launch, timed under "trace_closest"
        fused = sharded and int(st.context.get_option(L.OPT_PERSISTENT)) == 3 and info["depth"] <= 32 and \
            api.shard_num_samples(w, h, world) <= int(st.context.get_option(L.OPT_TAIL_PATHS)) // 4 * 3
        if fused:
            kname, kbytes, kms, klaunches = "k_tail_pool", bytes_closest + bytes_any, k_closest, tm["n_trace_closest"]
        elif k_closest >= k_any:
            kname, kbytes, kms, klaunches = variant % 0, bytes_closest, k_closest, tm["n_trace_closest"]
        else:
            any_name = "k_trace_static<2>" if int(st.context.get_option(L.OPT_PERSISTENT)) == 3 else variant % 2
            kname, kbytes, kms, klaunches = any_name, bytes_any, k_any, tm["n_trace_any"]
        launches_per_step = max(klaunches // args.steps, 1)
        kms_avg = kms / launches_per_step          # average duration of one launch of that kernel
        kbytes_launch = kbytes / launches_per_step
        achieved = kbytes_launch / (kms_avg * 1e-3) / 1e9 if kms_avg > 0 else 0.0
        out = {
            "metric": "Mray/s (primary+shadow), 1920x1080 spp=1, 1M-tri scene",
            "value": round(value, 3), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[3]: %d random triangles in the unit cube, %dx%d spp=%d, max depth %d "
                                   "(primary + NEE shadow ray), BVH2 %d nodes depth %d" % (
                                       args.tris, w, h, SPP, args.max_depth, info["num_nodes"], info["depth"]),
                       "rays_per_frame": rays_total, "closest_rays": counts["closest_rays"] if world == 1 else None,
                       "parallelism": "1 GPU" if world == 1 else "image tiles 8x4 interleaved over %d ranks + RCCL all-gather%s" % (
                           world, " (frame k's gather overlaps frame k+1's tracing)" if args.overlap else ""),
                       "bvh_build_s": round(info["build_seconds"], 3), "setup_s": round(build_s, 3)},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(kname) if world == 1 else None,
                         "algorithmic_bytes_per_launch": int(kbytes_launch),
                         "kernel_ms_avg": round(kms_avg, 4),
                         "per_frame_ms": {"trace_closest": round(k_closest, 4), "trace_any": round(k_any, 4),
                                          "shade": round(k_shade, 4), "raygen+resolve": round(k_other, 4)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(st, scene, w, h, args.max_depth, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    PT.cleanupState(st)


if __name__ == "__main__":
    main()
