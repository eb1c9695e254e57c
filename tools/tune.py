#!/usr/bin/env python3
"""Sweeps the traversal-kernel launch options on the headline workload and prints per-kernel times.
Usage (GPU box): python tools/tune.py [--tris N] [--depth D]"""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from optix_raytracer_amd import _lib as L  # noqa: E402
from optix_raytracer_amd import pathtracer as PT  # noqa: E402
from optix_raytracer_amd import scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--depth", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--grids", type=str, default="8,16,24,32")
    ap.add_argument("--stacks", type=str, default="16,32")
    ap.add_argument("--persistent", type=str, default="2,1,0")
    ap.add_argument("--refill", type=str, default="8")
    ap.add_argument("--leaf", type=str, default="1")
    ap.add_argument("--abandon", type=str, default="24")
    a = ap.parse_args()
    scene = scenes.random_soup(a.tris) if a.tris > 0 else scenes.cornell_box()
    st = PT.make_state(scene, a.width, a.height, 1, 0, max_depth=a.depth, log=lambda l, t, m: print('[%s] %s' % (t, m), flush=True), log_level=4)
    frame = torch.zeros((a.height, a.width, 4), dtype=torch.uint8, device=st.device)
    st.params.frame_buffer = frame.data_ptr()
    stream = torch.cuda.current_stream()
    print("accel:", st.context.accel_info(st.gas_handle))
    st.context.set_option(L.OPT_COUNT_VISITS, 1)
    st.context.launch(st.params, a.width, a.height, 1, stream=stream.cuda_stream)
    print("counts:", st.context.launch_stats())
    for refill, leaf, aband in ((8, 1, 0), (8, 1, 16), (8, 1, 32), (8, 1, 64)):
        st.context.set_option(L.OPT_REFILL_MIN, refill)
        st.context.set_option(L.OPT_LEAF_MIN, leaf)
        st.context.set_option(L.OPT_ABANDON_MIN, aband)
        st.context.launch(st.params, a.width, a.height, 1, stream=stream.cuda_stream)
        print("refill", refill, "leaf", leaf, "abandon", aband)
        st.context.launch_stats()
    st.context.set_option(L.OPT_COUNT_VISITS, 0)
    for pers, stack, grid, refill, leaf, aband in itertools.product([int(x) for x in a.persistent.split(",")],
                                                                    [int(x) for x in a.stacks.split(",")],
                                                                    [int(x) for x in a.grids.split(",")],
                                                                    [int(x) for x in a.refill.split(",")],
                                                                    [int(x) for x in a.leaf.split(",")],
                                                                    [int(x) for x in a.abandon.split(",")]):
        if pers != 2 and aband != int(a.abandon.split(",")[0]):
            continue
        st.context.set_option(L.OPT_ABANDON_MIN, aband)
        first = (stack == int(a.stacks.split(",")[0]) and grid == int(a.grids.split(",")[0]))
        first_thr = (refill == int(a.refill.split(",")[0]) and leaf == int(a.leaf.split(",")[0]))
        if pers == 0 and not (first and first_thr):
            continue
        if pers == 1 and (not first_thr or (stack == 32 and grid > 20)):
            continue
        if pers == 2 and stack != 16:
            continue
        st.context.set_option(L.OPT_REFILL_MIN, refill)
        st.context.set_option(L.OPT_LEAF_MIN, leaf)
        st.context.set_option(L.OPT_PERSISTENT, pers)
        st.context.set_option(L.OPT_LDS_STACK, stack)
        st.context.set_option(L.OPT_TRACE_GRID, grid)
        for _ in range(3):
            st.context.launch(st.params, a.width, a.height, 1, stream=stream.cuda_stream)
        st.context.set_option(L.OPT_PROFILE, 1)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(a.steps):
            st.context.launch(st.params, a.width, a.height, 1, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps * 1e3
        tm = st.context.launch_timings()
        st.context.set_option(L.OPT_PROFILE, 0)
        print("variant=%d lds_stack=%2d grid/CU=%2d refill=%2d leaf=%2d abandon=%2d : frame %.3f ms | closest %.3f any %.3f shade %.3f other %.3f" % (
            pers, stack, grid, refill, leaf, aband, dt, tm["ms_trace_closest"] / a.steps, tm["ms_trace_any"] / a.steps, tm["ms_shade"] / a.steps,
            (tm["ms_raygen"] + tm["ms_resolve"]) / a.steps), flush=True)
    PT.cleanupState(st)


if __name__ == "__main__":
    main()
