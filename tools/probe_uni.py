#!/usr/bin/env python3
"""How coherent are the lanes of a tile?  Renders the headline frame with a MIPT_PROBE_UNI build of the library and prints the
tiled kernel's per-launch sums (DeviceCounters::dbg): bulk steps, steps whose inner lanes stand at ONE node (the ones the scalar
fetch of MIPT_SCALAR_TOP serves), of those the ones with no lane at a leaf, the mean number of distinct nodes per step (counted up
to 8), steps with <= 2 and <= 4 distinct nodes.
usage (GPU box): make -C optix_raytracer_amd/csrc probe-uni PROBE_FLAGS=-DMIPT_PROBE_UNI=1; MIPT_LIBRARY=probe_uni.so python tools/probe_uni.py [W H]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MIPT_PROBE_PRINT"] = "1"
import torch  # noqa: E402,F401

from optix_raytracer_amd import pathtracer as PT  # noqa: E402
from optix_raytracer_amd import scenes  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
scene = scenes.random_soup(1_000_000)
st = PT.make_state(scene, W, H, 1, 0, max_depth=1)
ob = PT.OutputBuffer(W, H, st.device)
for i in range(3):
    PT.launchSubframe(ob, st)
    print("frame %d:" % i, flush=True)
    st.context.launch_stats()            # prints "probe dbg: ..." to stderr
