#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's headline configuration.

Workload (configs[3], SURVEY.md 8d "C4"): 1,000,000 random triangles in the unit cube, 1920x1080,
spp = 1, subframe 0, max depth 1 (the primary ray and its next-event-estimation shadow ray).
A "step" is one frame = one mipt_launch: raygen + closest hit (one kernel), shade (+ the pixels of the paths that end there),
any hit (+ the pixels whose last term it adds) -- three launches, DESIGN.md 3.1 / 3.3.  With N > 1 ranks
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
# ceilings, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0      # HBM3E 8.0 TB/s spec (:36)
L2_PEAK_GBS = 34500.0      # aggregate L2, 8 XCDs (:316)
CLOCK_HZ = 2.4e9           # max clock (:34)
N_CUS, N_SIMDS = 256, 1024
VALU_CYCLES = 2.0          # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles when other waves fill the SIMD (:54, :473)
VMEM_BASE_CYCLES, VMEM_LANE_CYCLES = 16.0, 0.48   # measured on this chip (profiles/r03_micro_vmem.txt): cycles a lane-divergent wave-level
                                                  # load costs its CU = 16 + 0.48 x lanes with an address of their own, any width
L1_PEAK_ACC_PER_CU_CLK = 1.0   # the vector L1 (TCP) looks up one cache line per cycle: a lane-divergent 16-byte load costs a
                               # lookup per lane (DESIGN.md 3.1 "what bounds it": measured 64 per wave instruction)


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
    ap.add_argument("--scatter-all", action="store_true",
                    help="N > 1: every rank un-permutes the gathered frame into the canonical layout (default: rank 0 only -- the rank that "
                         "shows the frame, as optixMultiGPU's GPU 0; the others hold the gathered float4 buffer)")
    ap.add_argument("--verify", action="store_true", help="debug: compare the gathered frame with a single-launch frame, bit for bit")
    ap.add_argument("--force-shard-path", action="store_true", help="debug: run the N > 1 code path (shard, gather, scatter) on one GPU")
    ap.add_argument("--grid", type=int, default=0, help="tuning: persistent workgroups per CU")
    ap.add_argument("--refill-min", type=int, default=0)
    ap.add_argument("--steal-min", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the bounded CPU-baseline sample")
    ap.add_argument("--builder", type=int, default=4,
                    help="MIPT_OPT_BUILD for the scene's BVH (the build is outside the timed region, as BASELINE.json's metric has it): 4 = device SAH with "
                         "spatial splits, the tree for a static scene (default here; what OPTIX_BUILD_FLAG_PREFER_FAST_TRACE asks for; ~12 ms per million "
                         "triangles); 3 = the same on the host (~3 s); 2 = the library's default, on the device in 7 ms; "
                         "1 = device Morton; 0 = host SAH.  The environment's MIPT_BUILD wins (A/B runs).")
    return ap.parse_args()


KERNEL_SOURCES = ("kernels.hip", "kernels.h", "pool_trace.h", "pool_core.h", "trav_core.h", "inst_pool.h", "inst_core.h", "wave_util.h", "detmath.h")


def kernels_blob_sha():
    """An id of the traversal kernels' source as it is in THIS tree: sha1 over the git blob ids (sha1 of "blob <len>\\0" + content) of
    kernels.hip and every header it includes, in KERNEL_SOURCES' order.  The GPU box has no .git, so the profile carries the id it was
    measured on (tools/pmc_json.py imports this function) and the two are compared here."""
    import hashlib
    h = hashlib.sha1()
    for name in KERNEL_SOURCES:
        data = open(os.path.join(ROOT, "optix_raytracer_amd", "csrc", name), "rb").read()
        h.update(hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest().encode())
    return h.hexdigest()


def pmc_counters(kernel):
    """Per-launch PMC counters of `kernel` from the COMMITTED rocprofv3 passes (profiles/pmc_kernels.json, written by
    tools/prof_l1.sh + tools/pmc_json.py).  They are properties of the kernel on this workload, not of this run: bench.py
    cannot collect them itself (that takes ~20 rocprofv3 passes).  The profile is used only when it was measured on the
    kernel sources of this tree (kernels_blob_sha(), recorded in the file) and names exactly this kernel: a profile of other kernel source is
    refused, not matched by name.  -> (counters or None, provenance)."""
    src = {"file": "profiles/pmc_kernels.json"}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_kernels.json")))
    except Exception:
        return None, dict(src, refused="no profile on file")
    src.update(commit=d.get("commit"), source=d.get("source"), kernels_hip_blob=d.get("kernels_hip_blob"))
    if d.get("kernels_hip_blob") != kernels_blob_sha():
        return None, dict(src, refused="the profile was measured on other kernel sources (id %s, this tree %s: kernels.hip and its headers): re-run tools/prof_l1.sh" % (
            str(d.get("kernels_hip_blob"))[:12], kernels_blob_sha()[:12]))
    if kernel not in d["kernels"]:
        return None, dict(src, refused="the profile holds no kernel named %s" % kernel)
    return d["kernels"][kernel], dict(src, kernel=kernel)


def ceilings(counters, t_s, alg_bytes):
    """The ceilings that can bind a traversal kernel, each as achieved / peak (MI355X_MICROARCH.md figures above):
    hbm   fabric bytes (2 * FETCH_SIZE + WRITE_SIZE: FETCH_SIZE doubled as the guide prescribes for gfx950) / t / 8 TB/s
    l2    TCC_REQ * 128 B / t / 34.5 TB/s
    l1    vector-L1 cache-line lookups per CU-cycle, TCP_TOTAL_CACHE_ACCESSES / (256 CUs * t * 2.4 GHz), peak 1
    ta    busy cycles of the texture addresser per CU-cycle (TA_TA_BUSY)
    td    TRANSFER cycles of the data-return unit per CU-cycle (TD_TD_BUSY - TD_TC_STALL; the stall share is reported beside it)
    vmem  wave-level divergent loads (TA_BUFFER_READ_WAVEFRONTS) * their measured cost in vector-memory pipeline cycles / (256 CUs * t * 2.4 GHz)
    valu  wave64 VALU instructions * 2 cycles / (1024 SIMDs * t * 2.4 GHz)
    Counter values come from the committed profile, the duration t from THIS run's HIP events."""
    c = {}
    if not counters or t_s <= 0:
        return c
    cyc = t_s * CLOCK_HZ
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        b = (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
        c["hbm"] = {"achieved": round(b / t_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b / t_s / 1e9 / HBM_PEAK_GBS, 4),
                    "bytes_per_launch": int(b)}
    if "TCC_REQ_sum" in counters:
        b = counters["TCC_REQ_sum"] * 128.0
        c["l2"] = {"achieved": round(b / t_s / 1e9, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(b / t_s / 1e9 / L2_PEAK_GBS, 4)}
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in counters:
        a = counters["TCP_TOTAL_CACHE_ACCESSES_sum"] / (N_CUS * cyc)
        c["l1"] = {"achieved": round(a, 4), "peak": L1_PEAK_ACC_PER_CU_CLK, "unit": "line lookups per CU-cycle", "frac": round(a / L1_PEAK_ACC_PER_CU_CLK, 4)}
    # the vector-memory pipeline of a CU: the texture addresser spends 16 cycles per dwordx4 wave instruction whatever the exec mask
    # (TA_BUFFER_TOTAL_CYCLES), the data-return unit is busy while it moves data AND while the load at its head waits for an L1 miss
    # (TD_TD_BUSY, in-order return)
    if "TA_TA_BUSY_sum" in counters:
        a = counters["TA_TA_BUSY_sum"] / (N_CUS * cyc)
        c["ta"] = {"achieved": round(a, 4), "peak": 1.0, "unit": "TA busy cycles per CU-cycle", "frac": round(a, 4)}
    if "TD_TD_BUSY_sum" in counters:
        # a THROUGHPUT ceiling counts the cycles the unit moves data, not the ones in which the load at the head of its in-order
        # queue waits for an L1 miss (TD_TC_STALL): a counter that calls waiting "busy" reads high on any latency-bound kernel
        stall = counters.get("TD_TC_STALL_sum", 0.0)
        a = (counters["TD_TD_BUSY_sum"] - stall) / (N_CUS * cyc)
        c["td"] = {"achieved": round(a, 4), "peak": 1.0, "unit": "TD transfer cycles per CU-cycle (TD_TD_BUSY - TD_TC_STALL)", "frac": round(a, 4),
                   "busy_incl_stall": round(counters["TD_TD_BUSY_sum"] / (N_CUS * cyc), 4),
                   "stall_share_of_busy": round(stall / counters["TD_TD_BUSY_sum"], 4) if counters["TD_TD_BUSY_sum"] else None}
    # The rate at which a CU takes lane-DIVERGENT loads (profiles/r03_micro_vmem.txt, tools/micro/tcp_*.hip on this chip): a wave-level
    # buffer load costs its CU's vector-memory pipeline ~16 + 0.48 x (lanes with an address of their own) cycles whatever its width
    # (46.8 with 64 such lanes), L1-resident data.  `frac` prices each load at the kernel's lane activity (VALU lane utilisation as the
    # proxy), `frac_if_every_lane_diverges` at 46.8 cycles.
    if "TA_BUFFER_READ_WAVEFRONTS_sum" in counters:
        loads = counters["TA_BUFFER_READ_WAVEFRONTS_sum"]
        util = 1.0
        if counters.get("SQ_THREAD_CYCLES_VALU") and counters.get("SQ_ACTIVE_INST_VALU"):
            util = min(1.0, counters["SQ_THREAD_CYCLES_VALU"] / (counters["SQ_ACTIVE_INST_VALU"] * 64.0))
        cost = VMEM_BASE_CYCLES + VMEM_LANE_CYCLES * 64.0 * util
        a = loads * cost / (N_CUS * cyc)
        c["vmem"] = {"achieved": round(a, 4), "peak": 1.0, "unit": "vector-memory pipeline cycles per CU-cycle (wave-level divergent loads x measured cost)",
                     "frac": round(a, 4), "frac_if_every_lane_diverges": round(loads * (VMEM_BASE_CYCLES + VMEM_LANE_CYCLES * 64.0) / (N_CUS * cyc), 4),
                     "wave_loads_per_launch": int(loads), "cycles_per_wave_load": round(cost, 1), "lane_activity": round(util, 3)}
    if "SQ_INSTS_VALU" in counters:
        a = counters["SQ_INSTS_VALU"] * VALU_CYCLES / (N_SIMDS * cyc)
        c["valu"] = {"achieved": round(a, 4), "peak": 1.0, "unit": "SIMD issue cycles per cycle", "frac": round(a, 4)}
    return c


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


def physical_cores(limit):
    """One logical CPU per physical core of this process's affinity mask (first hardware thread of each core), at most
    `limit` of them: the CPUs the baseline's threads are pinned to."""
    seen, picked = set(), []
    for cpu in sorted(os.sched_getaffinity(0)):
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu).read().strip()
        except Exception:
            sib = str(cpu)
        if sib in seen:
            continue
        seen.add(sib)
        picked.append(cpu)
    return picked[:max(1, limit)]


def build_cpubench():
    """libcpubench.so is -march=native code: built on the host it runs on, under a name that carries that host's CPU."""
    import hashlib
    import subprocess
    try:
        info = open("/proc/cpuinfo").read()
        key = "".join(l for l in info.splitlines(True) if l.startswith(("model name", "flags")))[:8192]
    except Exception:
        key = "unknown"
    name = "libcpubench.%s.so" % hashlib.sha1(key.encode()).hexdigest()[:10]
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), name, "BENCH_OUT=" + name], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "oracle", name)


def cpu_baseline(st, scene, width, height, max_depth, budget_s):
    """The CPU oracle (oracle/pt_oracle.c, kind "port": this build's C restatement; the reference has no CPU path)
    rendering the same frame over the same BVH: oracle/cpu_bench.c -- built -O3 -march=native -fno-fast-math
    -ffp-contract=off on this host -- runs one PINNED native thread per physical core (at most the cgroup's CPU quota),
    4-row bands from an atomic counter, three repetitions of budget_s / 3, median rate."""
    import ctypes as C

    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from optix_raytracer_amd import scenes

    class OrcBenchResult(C.Structure):
        _fields_ = [("mray_per_s", C.c_double), ("rep_mray_per_s", C.c_double * 3), ("seconds", C.c_double), ("frames", C.c_double),
                    ("rays", C.c_uint64), ("threads", C.c_int32), ("pinned", C.c_int32)]
    lib = C.CDLL(build_cpubench())
    lib.orc_scene_create.restype = C.c_void_p
    lib.orc_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.orc_scene_set_bvh.restype = C.c_int
    lib.orc_scene_set_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    lib.orc_scene_destroy.argtypes = [C.c_void_p]
    lib.orc_bench_render.restype = C.c_int
    lib.orc_bench_render.argtypes = [C.c_void_p, C.POINTER(O.OrcParams), C.c_uint32, C.c_int, C.POINTER(C.c_int), C.c_double,
                                     C.POINTER(OrcBenchResult)]
    nodes, tris = st.context.accel_export(st.gas_handle)
    nodes, tris = np.ascontiguousarray(nodes), np.ascontiguousarray(tris)
    verts = np.ascontiguousarray(scene["vertices"], np.float32).reshape(-1, 4)
    mat = np.ascontiguousarray(scene["mat_index"], np.uint32)
    em = np.ascontiguousarray(scene["emission"], np.float32).reshape(-1, 3)
    df = np.ascontiguousarray(scene["diffuse"], np.float32).reshape(-1, 3)
    bg = np.ascontiguousarray(scene.get("bg", np.zeros(4)), np.float32)
    h = lib.orc_scene_create(verts.ctypes.data, verts.shape[0] // 3, mat.ctypes.data, em.ctypes.data, df.ctypes.data, em.shape[0], bg.ctypes.data)
    if lib.orc_scene_set_bvh(h, nodes.ctypes.data, nodes.nbytes // 64, tris.ctypes.data, tris.nbytes // 48):
        raise SystemExit("cpu_baseline: the exported BVH failed the oracle's validation")
    launch = scenes.launch_desc(scene, width, height, SPP)
    accum = np.zeros((height, width, 4), np.float32)
    p = O.Scene.make_params(None, launch, accum, None)
    cpus = physical_cores(effective_cpus())
    ids = (C.c_int * len(cpus))(*cpus)
    res = OrcBenchResult()
    rc = lib.orc_bench_render(h, C.byref(p), max_depth, len(cpus), ids, float(budget_s), C.byref(res))
    lib.orc_scene_destroy(h)
    if rc:
        raise SystemExit("cpu_baseline: orc_bench_render failed (%d)" % rc)
    return {"value": round(res.mray_per_s, 3), "unit": "Mray/s", "cores": len(cpus), "kind": "port",
            "per_core": round(res.mray_per_s / len(cpus), 3),
            "sample": "%.1f frames of the same %dx%d workload (%d rays) in %.1f s wall: %d native threads pinned one per physical core "
                      "(%d pinned), 4-row bands from an atomic counter, median of 3 repetitions (%s Mray/s), gcc -O3 -march=native "
                      "-fno-fast-math -ffp-contract=off" % (
                          res.frames, width, height, res.rays, res.seconds, len(cpus), res.pinned,
                          " / ".join("%.2f" % v for v in res.rep_mray_per_s))}


def spawn_ranks(args):
    """`python bench.py --gpus N` started plainly: start the N ranks as CHILD processes (torch.distributed.run, one per GPU)
    before this process has touched the GPU, let rank 0's JSON line through on the inherited stdout, exit with the
    children's code.  (Never exec from a process that has initialised HIP.)"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    from optix_raytracer_amd import _lib as L
    from optix_raytracer_amd import api, scenes
    from optix_raytracer_amd import dist as D
    from optix_raytracer_amd import pathtracer as PT

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    os.environ.setdefault("MIPT_BUILD", str(args.builder))          # read when the context is created
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
            if rank == 0 or args.scatter_all:
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
    # ---- timed region: exactly K steps between barrier + synchronize.  HIP events bracket the two traversal kernels on the
    # launch stream in every 8th step (MIPT_OPT_PROFILE 3: an event record costs ~4 us of stream time -- four per frame are
    # 2.6 % of it, bracketing all kernels 4 %; the sample starts with the first timed step)
    st.context.set_option(L.OPT_PROFILE, 3)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    tm = st.context.launch_timings()              # event sums over all K timed steps (read after the fence)
    sampled_steps = (args.steps + 7) // 8
    k_closest, k_any = (tm[k] / sampled_steps for k in ("ms_trace_closest", "ms_trace_any"))
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

    # ---- the first frame after a change (camera size, rank, option): no cost order of the tiles exists yet and the primary launch
    # walks them in image order.  The timed region above is the steady state of a progressive renderer (identical frames, perfect
    # order); this is the other end.  Setting MIPT_OPT_TILE_ORDER discards the order; median of 5 such frames, each fenced.
    first_ms = []
    for _ in range(5):
        st.context.set_option(L.OPT_TILE_ORDER, int(st.context.get_option(L.OPT_TILE_ORDER)))
        fence()
        t1 = time.perf_counter()
        step()
        fence()
        first_ms.append((time.perf_counter() - t1) * 1e3)
    first_frame_ms = sorted(first_ms)[len(first_ms) // 2]
    for _ in range(2):
        step()                               # leave the context in its steady state again
    fence()

    # ---- the same frames over the tree the library's DEFAULT builder makes (MIPT_OPT_BUILD = 2, on the device), beside the headline number:
    # `value` is measured over the tree --builder asks for (4: spatial splits); one GPU, plain launches only
    default_builder = None
    if world == 1 and not sharded and os.environ.get("MIPT_BUILD") != "2":
        keep_env = os.environ.get("MIPT_BUILD")
        os.environ["MIPT_BUILD"] = "2"
        try:
            st2 = PT.make_state(scene, w, h, SPP, local_rank, max_depth=args.max_depth)
            info2 = st2.context.accel_info(st2.gas_handle)
            frame2 = torch.zeros((h, w, 4), dtype=torch.uint8, device=st2.device)
            st2.params.frame_buffer = frame2.data_ptr(); st2.params.subframe_index = 0
            torch.cuda.synchronize()
            for _ in range(max(args.warmup, 5)):
                st2.context.launch(st2.params, w, h, 1, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            n2 = max(args.steps, 20)
            t2 = time.perf_counter()
            for _ in range(n2):
                st2.context.launch(st2.params, w, h, 1, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t2) / n2 * 1e3
            default_builder = {"ms_per_step": round(ms2, 4), "bvh_build_s": round(info2["build_seconds"], 4),
                               "tree": "BVH2 %d nodes depth %d over %d triangle records (MIPT_OPT_BUILD = 2: device binned SAH + SAH treelets)" % (
                                   info2["num_nodes"], info2["depth"], info2["num_triangles"]),
                               "identical_frame": bool(torch.equal(frame2, frame) and torch.equal(st2.accum, st.accum))}
            PT.cleanupState(st2)
        finally:
            if keep_env is None:
                os.environ.pop("MIPT_BUILD", None)
            else:
                os.environ["MIPT_BUILD"] = keep_env

    # ---- N > 1 (or --force-shard-path): what a rank's step consists of -- render (mipt_launch_shard on the launch stream), gather (the
    # all-gather on the exchange stream), scatter (mipt_unpermute) -- from events in a few extra, untimed steps; max over ranks below
    per_rank = None
    if sharded:
        ne = 8
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(ne)]
        for i in range(ne):
            b = state["k"] & 1
            state["k"] += 1
            stream.wait_event(consumed[b])
            ev[i][0].record(stream)
            st.context.launch_shard(st.params, w, h, rank, world, compact[b].data_ptr(), stream=stream.cuda_stream)
            ev[i][1].record(stream)
            rendered[b].record(stream)
            with torch.cuda.stream(xstream):
                xstream.wait_event(rendered[b])
                ev[i][2].record(xstream)
                if world == 1:
                    gathered[b].copy_(compact[b])
                elif args.backend == "nccl":
                    dist.all_gather_into_tensor(gathered[b], compact[b])
                else:
                    gathered[b].copy_(D.all_gather_shards(compact[b].cpu(), world))
                ev[i][3].record(xstream)
                consumed[b].record(xstream)
                ev[i][4].record(xstream)
                if rank == 0 or args.scatter_all:
                    st.context.unpermute(gathered[b].data_ptr(), world, w, h, st.accum.data_ptr(), frame.data_ptr(), stream=xstream.cuda_stream)
                ev[i][5].record(xstream)
        fence()
        per_rank = [sum(e[2 * j].elapsed_time(e[2 * j + 1]) for e in ev) / ne for j in range(3)]

    if args.verify and sharded and (rank == 0 or args.scatter_all):
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
        pr = torch.tensor(per_rank, dtype=torch.float64, device=red_dev)
        dist.all_reduce(pr, op=dist.ReduceOp.MAX)
        per_rank = [float(x) for x in pr.tolist()]

    if rank == 0:
        ms_per_step = t_max / args.steps * 1e3
        value = rays_total * args.steps / t_max / 1e6
        # ---- roofline of the dominant kernel (algorithmic bytes: SURVEY.md 8d / DESIGN.md)
        nv_c = counts["node_visits"] - counts["shadow_node_visits"]
        tt_c = counts["tri_tests"] - counts["shadow_tri_tests"]
        bytes_closest = 64 * nv_c + 48 * tt_c + (32 + 16) * counts["closest_rays"]
        bytes_any = 64 * counts["shadow_node_visits"] + 48 * counts["shadow_tri_tests"] + (32 + 16 + 32) * counts["shadow_rays"]
        q = str(int(info.get("node_format", 0)))
        # (the tiled kernel makes the primary rays itself -- its last template argument -- unless MIPT_FUSE_RAYGEN=0)
        rg = "true" if os.environ.get("MIPT_FUSE_RAYGEN", "1") != "0" else "false"
        variant = {3: "k_trace_pool<%%d, %d, %s, %s>" % (24 if info["depth"] <= 24 else 32, "1" if q != "0" else "0", rg), 2: "k_trace_stream<%d, false>",
                   0: "k_trace_simple<%d, false>"}[int(st.context.get_option(L.OPT_PERSISTENT))]
        # a pass of at most MIPT_OPT_TAIL_PATHS paths (a rank's shard at N >= 2) runs in the path kernel: closest-hit and
        # any-hit traversal, shading and accumulation of a wave's 64 paths in ONE launch, timed under "trace_closest"
        fused = sharded and int(st.context.get_option(L.OPT_PERSISTENT)) == 3 and info["depth"] <= 32 and \
            api.shard_num_samples(w, h, world) <= int(st.context.get_option(L.OPT_TAIL_PATHS)) // 4 * 3
        if fused:
            kname, kbytes, kms, klaunches = "k_tail_pool<%s>" % q, bytes_closest + bytes_any, k_closest, tm["n_trace_closest"]
        elif k_closest >= k_any:
            kname, kbytes, kms, klaunches = variant % 0, bytes_closest, k_closest, tm["n_trace_closest"]
        else:
            any_name = "k_trace_static<2, %s>" % q if int(st.context.get_option(L.OPT_PERSISTENT)) == 3 else variant % 2
            kname, kbytes, kms, klaunches = any_name, bytes_any, k_any, tm["n_trace_any"]
        launches_per_step = max(klaunches // sampled_steps, 1)
        kms_avg = kms / launches_per_step          # average duration of one launch of that kernel
        kbytes_launch = kbytes / launches_per_step
        alg_gbs = kbytes_launch / (kms_avg * 1e-3) / 1e9 if kms_avg > 0 else 0.0
        # ---- the ceilings that can bind: HBM (counter bytes), L2, vector L1, VALU issue.  The BVH (~115 MB) is resident in L2 /
        # Infinity Cache, so algorithmic bytes / t exceeds what HBM could deliver and is reported as information only.
        counters, counters_from = pmc_counters(kname) if world == 1 else (None, {"refused": "N > 1: the profile is of the one-GPU kernels"})
        ceil = ceilings(counters, kms_avg * 1e-3, kbytes_launch)
        # SURVEY.md 8(d)'s number: HBM bytes (PMC counters, per launch) / the kernel's live duration / 8 TB/s.  `bound` says what the
        # kernel is limited by: the largest THROUGHPUT ceiling when one exceeds 0.7, else "latency" -- no pipe is saturated and the
        # waves wait for dependent fetches; the stall share of the data-return unit's busy time is reported with it.
        hbm = ceil.get("hbm", {"achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "bytes_per_launch": None})
        thr = {k: v["frac"] for k, v in ceil.items()}
        bound, limiter = "hbm", None
        if thr:
            top = max(thr, key=lambda k: thr[k])
            if thr[top] > 1.0:
                counters_from = dict(counters_from or {}, stale="a ceiling computed from these counters exceeds 1 for this run's kernel time: re-profile")
            bound = top if thr[top] >= 0.7 else "latency"
            limiter = {"largest_throughput_ceiling": top, "its_frac": thr[top],
                       "td_stall_share_of_busy": ceil.get("td", {}).get("stall_share_of_busy"),
                       "reading": ("no throughput ceiling above 0.7: the kernel waits for dependent fetches; the nearest ceiling is %s" % top)
                                  if bound == "latency" else "bound by the %s ceiling" % top}
        out = {
            "metric": "Mray/s (primary+shadow), 1920x1080 spp=1, 1M-tri scene",
            "value": round(value, 3), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[3]: %d random triangles in the unit cube, %dx%d spp=%d, max depth %d "
                                   "(primary + NEE shadow ray), BVH2 %d nodes depth %d over %d triangle records (builder %s: %s), node format %d (%d mirror nodes, stack %d)" % (
                                       args.tris, w, h, SPP, args.max_depth, info["num_nodes"], info["depth"], info["num_triangles"], os.environ.get("MIPT_BUILD"),
                                       {"0": "host binned SAH", "1": "device Morton + SAH treelets", "2": "device binned SAH + SAH treelets",
                                        "3": "host binned SAH with spatial splits", "4": "device binned SAH with spatial splits + SAH treelets"}.get(os.environ.get("MIPT_BUILD"), "?"),
                                       info["node_format"], info["mirror_nodes"], info["stack_entries"]),
                       "rays_per_frame": rays_total, "closest_rays": counts["closest_rays"] if world == 1 else None,
                       "parallelism": "1 GPU" if world == 1 else "image tiles 8x4 interleaved over %d ranks + RCCL all-gather%s" % (
                           world, " (frame k's gather overlaps frame k+1's tracing)" if args.overlap else ""),
                       "bvh_build_s": round(info["build_seconds"], 3), "setup_s": round(build_s, 3),
                       "default_builder": default_builder,
                       "first_frame_ms": round(first_frame_ms, 4),
                       "first_frame_is": "median of 5 fenced frames launched with no tile cost order (image order), as after any change of "
                                         "camera size, rank or option; ms_per_step is the steady state of identical progressive frames"},
            "roofline": {"bound": bound, "kernel": kname, "achieved": hbm["achieved"], "peak": hbm["peak"], "unit": hbm["unit"],
                         "frac": hbm["frac"], "frac_is": "HBM bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE) / kernel time / 8 TB/s",
                         "traffic": hbm.get("bytes_per_launch"),
                         "limiter": limiter, "ceilings": ceil, "counters_from": counters_from,
                         "algorithmic_bytes_per_launch": int(kbytes_launch), "algorithmic_GBps_informational": round(alg_gbs, 1),
                         "kernel_ms_avg": round(kms_avg, 4), "kernel_ms_from": "HIP events around the kernel in every 8th timed step (%d launches)" % klaunches,
                         "per_frame_ms": {"trace_closest": round(k_closest, 4), "trace_any": round(k_any, 4),
                                          "shade": round(k_shade, 4), "raygen+resolve": round(k_other, 4)}},
        }
        if sharded:
            out["per_rank_ms"] = {"render": round(per_rank[0], 4), "gather": round(per_rank[1], 4), "scatter": round(per_rank[2], 4),
                                  "is": "max over ranks of the mean over 8 extra untimed steps (events on the launch / exchange streams); "
                                        "gather and scatter of frame k run beside the render of frame k+1%s; the scatter into the canonical layout "
                                        "runs on %s" % ("" if args.overlap else " (not in this run: --no-overlap)", "every rank" if args.scatter_all else "rank 0 only"),
                                  "ranks": world, "backend": ("rccl (torch.distributed nccl), %d ranks" % dist.get_world_size()) if world > 1 and args.backend == "nccl"
                                  else (args.backend if world > 1 else "none (one rank: device copy)")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(st, scene, w, h, args.max_depth, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    PT.cleanupState(st)


if __name__ == "__main__":
    main()
