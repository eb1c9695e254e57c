#!/usr/bin/env python3
"""Experiment: what would a hipGraph of the frame's three launches save?  One headline frame captured on the caller's stream
(mipt_launch is capturable once its buffers exist: it only launches kernels then) and replayed, against the same frame launched
normally, back to back."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optix_raytracer_amd import pathtracer as PT, scenes

w, h = 1920, 1080
scene = scenes.random_soup(1_000_000)
st = PT.make_state(scene, w, h, 1, 0, max_depth=1)
ob = PT.OutputBuffer(w, h, st.device)
st.params.frame_buffer = ob.map()
stream = st.stream


def launch():
    st.context.launch(st.params, w, h, 1, stream=stream.cuda_stream)


for _ in range(20):
    launch()
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    launch()
torch.cuda.synchronize()
print("stream launches: %.4f ms per frame" % ((time.perf_counter() - t0) / N * 1e3), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=stream):
    launch()
torch.cuda.synchronize()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    g.replay()
torch.cuda.synchronize()
print("graph replays:   %.4f ms per frame (one frame per graph)" % ((time.perf_counter() - t0) / N * 1e3), flush=True)
g4 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g4, stream=stream):
    for _ in range(4):
        launch()
torch.cuda.synchronize()
for _ in range(5):
    g4.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N // 4):
    g4.replay()
torch.cuda.synchronize()
print("graph replays:   %.4f ms per frame (four frames per graph)" % ((time.perf_counter() - t0) / (N // 4 * 4) * 1e3), flush=True)
