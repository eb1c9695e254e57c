#!/usr/bin/env python3
"""Sweeps the traversal-kernel launch options on the headline workload and prints per-kernel times.
Usage (GPU box): python tools/tune.py [--tris N] [--depth D] [--variants 2,0] [--grids ..] [--refill ..] [--steal ..]"""
import argparse
import itertools
import os
import sys
import time

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
    ap.add_argument("--variants", type=str, default="2,0")
    ap.add_argument("--grids", type=str, default="24")
    ap.add_argument("--refill", type=str, default="8")
    ap.add_argument("--steal", type=str, default="32")
    ap.add_argument("--burst", type=str, default="0x0808")
    a = ap.parse_args()
    ints = lambda s: [int(x) for x in s.split(",")]
    scene = scenes.random_soup(a.tris) if a.tris > 0 else scenes.cornell_box()
    st = PT.make_state(scene, a.width, a.height, 1, 0, max_depth=a.depth)
    frame = torch.zeros((a.height, a.width, 4), dtype=torch.uint8, device=st.device)
    st.params.frame_buffer = frame.data_ptr()
    stream = torch.cuda.current_stream()
    print("accel:", st.context.accel_info(st.gas_handle))
    st.context.set_option(L.OPT_COUNT_VISITS, 1)
    st.context.launch(st.params, a.width, a.height, 1, stream=stream.cuda_stream)
    print("counts:", st.context.launch_stats())
    st.context.set_option(L.OPT_COUNT_VISITS, 0)
    for variant in ints(a.variants):
        bursts = [int(x, 0) for x in a.burst.split(",")]
        combos = itertools.product(ints(a.grids), ints(a.refill), ints(a.steal), bursts) if variant >= 2 else [(ints(a.grids)[0], 8, 32, 0x0101)]
        for grid, refill, steal, burst in combos:
            st.context.set_option(L.OPT_BURST, burst)
            st.context.set_option(L.OPT_PERSISTENT, variant)
            st.context.set_option(L.OPT_TRACE_GRID, grid if variant else 0)
            st.context.set_option(L.OPT_REFILL_MIN, refill)
            st.context.set_option(L.OPT_STEAL_MIN, steal)
            for _ in range(3):
                st.context.launch(st.params, a.width, a.height, 1, stream=stream.cuda_stream)
            st.context.set_option(L.OPT_PROFILE, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                st.context.launch(st.params, a.width, a.height, 1, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps * 1e3
            tm = st.context.launch_timings()
            st.context.set_option(L.OPT_PROFILE, 0)
            print("variant=%d grid/CU=%2d refill=%2d steal=%2d burst=0x%04x : frame %.3f ms | closest %.3f any %.3f shade %.3f other %.3f" % (
                variant, grid, refill, steal, burst, dt, tm["ms_trace_closest"] / a.steps, tm["ms_trace_any"] / a.steps,
                tm["ms_shade"] / a.steps, (tm["ms_raygen"] + tm["ms_resolve"]) / a.steps), flush=True)
    PT.cleanupState(st)


if __name__ == "__main__":
    main()
