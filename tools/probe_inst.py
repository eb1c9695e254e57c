#!/usr/bin/env python3
"""Work counters of the two-level entry pool (probe build: make -C optix_raytracer_amd/csrc probe-inststats PROBE_FLAGS=-DMIPT_INST_STATS=1,
run with MIPT_LIBRARY=probe_inststats.so): per launch, rounds / live lanes / top-level nodes / instance nodes / triangles / pool fill,
for the primary rays and the shadow rays of the instanced MeshViewer scene."""
import os, pathlib, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ctypes as C
from optix_raytracer_amd import meshviewer as MV, scenes, _lib as L
tmp = pathlib.Path(tempfile.mkdtemp())
for n_inst in [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else ["150", "2000"])]:
    path = scenes.write_instanced_gltf(tmp / ("inst%d.gltf" % n_inst), n_instances=n_inst, seed=3)
    st, img = MV.render(path, 1920, 1080, subframes=1, two_level=True)
    out = MV.OutputBuffer(1920, 1080, st.device)
    ctx = st.context
    # the counters sit in DeviceCounters::dbg; read them around one subframe
    def dbg():
        os.environ["MIPT_PROBE_PRINT"] = "1"
        r, w = os.pipe(); old = os.dup(2); os.dup2(w, 2)
        ctx.launch_stats()
        os.dup2(old, 2); os.close(w)
        txt = os.read(r, 4096).decode(); os.close(r)
        return np.array([int(x) for x in txt.split("probe dbg:")[1].split()], np.int64)
    a = dbg()
    MV.updateState(st); MV.launchSubframe(out, st)
    b = dbg()
    d = b - a
    print("%d instances: rounds %d, live lanes per round %.1f, entries: top-level nodes %d, instance nodes %d, triangles %d; mean pool fill %.0f" % (
        n_inst, d[0], d[1] / max(d[0], 1), d[2], d[3], d[4], d[5] / max(d[0], 1)), flush=True)
    s = ctx.launch_stats()
    print("      rays: %d closest + %d shadow -> per ray %.1f top-level nodes, %.1f instance nodes, %.1f triangles" % (
        s["closest_rays"], s["shadow_rays"], d[2] / (s["closest_rays"] + s["shadow_rays"]), d[3] / (s["closest_rays"] + s["shadow_rays"]), d[4] / (s["closest_rays"] + s["shadow_rays"])))
    MV.cleanup(st)
