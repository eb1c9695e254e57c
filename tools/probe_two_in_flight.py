#!/usr/bin/env python3
"""Throughput of the headline frame with N progressive renders in flight on one GPU: N contexts (each with its own queues, accumulators and
tree) on N streams, launched in turn.  The kernels of a frame end in tails with idle wave slots; frames of other streams can
fill them.  usage: probe_two_in_flight.py [n_contexts ...]   (default 1 2 3)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIPT_BUILD", "4")
import torch
from optix_raytracer_amd import pathtracer as PT, scenes
W, H, K = 1920, 1080, 120
scene = scenes.random_soup(1_000_000)
for n in [int(x) for x in sys.argv[1:]] or [1, 2, 3]:
    sts = [PT.make_state(scene, W, H, 1, 0, max_depth=1) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    frames = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
    for st, f in zip(sts, frames):
        st.params.frame_buffer = f.data_ptr(); st.params.subframe_index = 0
    def step(i):
        st = sts[i % n]
        st.context.launch(st.params, W, H, 1, stream=streams[i % n].cuda_stream)
    for i in range(12 * n):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    rays = sts[0].context.launch_stats()
    per = rays["closest_rays"] + rays["shadow_rays"]
    print("%d in flight: %.4f ms per frame, %.1f Mray/s" % (n, ms, per / ms / 1e3), flush=True)
    for st in sts:
        PT.cleanupState(st)
