#!/usr/bin/env python3
"""Measures the per-step latency of a traversal: times mipt_trace_closest on a few LONG rays only
(found with the CPU oracle's step profile), for the kernel variants."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import _oracle as O  # noqa: E402
from optix_raytracer_amd import _lib as L  # noqa: E402
from optix_raytracer_amd import pathtracer as PT  # noqa: E402
from optix_raytracer_amd import scenes  # noqa: E402

scene = scenes.random_soup(1_000_000)
W, H = 1920, 1080
st = PT.make_state(scene, W, H, 1, 0, max_depth=1)
nodes, tris = st.context.accel_export(st.gas_handle)
osc = O.Scene(scene).set_bvh(nodes, tris)
launch = scenes.launch_desc(scene, W, H)
lib = O.lib()
lib.orc_trace_profile.restype = C.c_uint32
lib.orc_trace_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
buf = np.zeros(2 * 4096, np.uint16)
cand = []
for y in range(0, H, 6):
    rows = osc.primary_rays(launch, rows=(y, y + 1))
    for x in range(0, W, 3):
        r = np.ascontiguousarray(rows[x])
        n = lib.orc_trace_profile(osc._h, r.ctypes.data, 0, buf.ctypes.data, 4096)
        p = buf[:2 * min(n, 4096)].reshape(-1, 2)
        cand.append((int(p[:, 0].sum()), int(p[p[:, 1] != 0xffff, 1].sum()), r.copy()))
cand.sort(key=lambda c: -(c[0] + c[1]))
print("longest rays (nodes, tris):", [(c[0], c[1]) for c in cand[:8]])
st.context.set_option(L.OPT_PROFILE, 1)
for variant in (0, 2):
    st.context.set_option(L.OPT_PERSISTENT, variant)
    for label, rays in (("1 longest ray", np.stack([cand[0][2]])),
                        ("64 x longest ray", np.stack([cand[0][2]] * 64)),
                        ("64 longest rays", np.stack([c[2] for c in cand[:64]])),
                        ("4096 longest rays", np.stack([c[2] for c in cand[:4096]])),
                        ("64 median rays", np.stack([c[2] for c in cand[len(cand) // 4:len(cand) // 4 + 64]]))):
        d = torch.from_numpy(np.ascontiguousarray(rays, np.float32)).cuda()
        hits = torch.zeros((rays.shape[0], 4), dtype=torch.float32, device="cuda")
        for _ in range(3):
            st.context.trace_closest(st.gas_handle, d.data_ptr(), rays.shape[0], hits.data_ptr())
        st.context.launch_timings()
        reps = 10
        for _ in range(reps):
            st.context.trace_closest(st.gas_handle, d.data_ptr(), rays.shape[0], hits.data_ptr())
        tm = st.context.launch_timings()
        ms = tm["ms_trace_closest"] / reps
        steps = cand[0][0] + cand[0][1] if "longest ray" in label else None
        print("variant %d  %-18s : %.4f ms%s" % (variant, label, ms, ("  -> %.3f us/step" % (ms * 1e3 / steps)) if steps else ""), flush=True)
PT.cleanupState(st)
