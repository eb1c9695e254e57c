#!/usr/bin/env python3
"""Experiment: how much of the frame time is idle capacity that ANOTHER frame's kernels could fill?  Two contexts render the
headline frame on two streams at once (their kernels interleave on the device: one's closest-hit launch runs beside the
other's shade / any-hit launches and their drains); aggregate frames per second against one context alone."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optix_raytracer_amd import pathtracer as PT, scenes

w, h = 1920, 1080
scene = scenes.random_soup(1_000_000)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
states = []
for _ in range(K):
    st = PT.make_state(scene, w, h, 1, 0, max_depth=1)
    ob = PT.OutputBuffer(w, h, st.device)
    states.append((st, ob))


def launch(st, ob):
    st.params.frame_buffer = ob.map()
    st.context.launch(st.params, w, h, 1, stream=st.stream.cuda_stream)


for st, ob in states:
    for _ in range(10):
        launch(st, ob)
torch.cuda.synchronize()
N = 200
for k in range(1, K + 1):
    t0 = time.perf_counter()
    for _ in range(N):
        for st, ob in states[:k]:
            launch(st, ob)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d context(s) at once: %.4f ms per frame (%.4f ms per round of %d)" % (k, dt / (N * k) * 1e3, dt / N * 1e3, k), flush=True)
