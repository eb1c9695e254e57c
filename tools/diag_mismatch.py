#!/usr/bin/env python3
"""Renders the headline frame on the GPU and with the oracle and reports differing pixels."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _oracle as O
from optix_raytracer_amd import _lib as L, pathtracer as PT, scenes
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scene = scenes.random_soup(n_tris)
W, H = 1920, 1080
st = PT.make_state(scene, W, H, 1, 0, max_depth=1)
ob = PT.OutputBuffer(W, H, st.device)
nodes, tris = st.context.accel_export(st.gas_handle)
osc = O.Scene(scene).set_bvh(nodes, tris)
a, f, ost = osc.render(scenes.launch_desc(scene, W, H, 1), max_depth=1, threads=16)
for steal in (65, 32, 8, 1):
    st.context.set_option(L.OPT_STEAL_MIN, steal)
    for rep in range(3):
        PT.launchSubframe(ob, st)
        acc = st.accum.cpu().numpy()
        bad = np.argwhere((acc.view(np.uint32) != a.view(np.uint32)).any(axis=2))
        print("steal", steal, "rep", rep, "differing pixels:", len(bad), [(tuple(b), acc[tuple(b)][:3], a[tuple(b)][:3]) for b in bad[:3]], flush=True)
