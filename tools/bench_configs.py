#!/usr/bin/env python3
"""Times the BASELINE.json configurations that fit one GPU (besides the headline one that bench.py owns) and
prints one line each: ms per launch, Mray/s (closest + shadow rays actually traced)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optix_raytracer_amd import _lib as L, pathtracer as PT, scenes

ONLY = sys.argv[1:]          # optional name prefixes ("C1", "C2" ...): run only these

def run(name, scene, w, h, spp, max_depth, reps, batch=0):
    if ONLY and not any(name.startswith(o) for o in ONLY): return
    scene = scene()
    st = PT.make_state(scene, w, h, spp, 0, max_depth=max_depth)
    st.context.set_option(L.OPT_BATCH_SAMPLES, batch)
    ob = PT.OutputBuffer(w, h, st.device)
    PT.launchSubframe(ob, st)
    stats = st.context.launch_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        PT.launchSubframe(ob, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    rays = stats["closest_rays"] + stats["shadow_rays"]
    print("%-44s %9.3f ms/launch  %9.1f Mray/s  (%d closest + %d shadow rays, %d wavefront iterations)" % (
        name, ms, rays / ms / 1e3, stats["closest_rays"], stats["shadow_rays"], stats["bounces"]), flush=True)
    PT.cleanupState(st)

run("C1 Cornell 512x512 spp1 (RR depth)", scenes.cornell_box, 512, 512, 1, 0, 5)
run("C2 Cornell 1920x1080 spp16 (RR depth)", scenes.cornell_box, 1920, 1080, 16, 0, 2)
run("C2 Cornell 1920x1080 spp16 (RR), batched samples", scenes.cornell_box, 1920, 1080, 16, 0, 3, batch=1)
run("C4 soup 1M 1920x1080 spp1 depth 1", lambda: scenes.random_soup(1_000_000), 1920, 1080, 1, 1, 10)
run("C4 soup 1M 1920x1080 spp1 depth 8", lambda: scenes.random_soup(1_000_000), 1920, 1080, 1, 8, 5)
run("S1 soup 1M 1920x1080 spp1 (RR depth)", lambda: scenes.random_soup(1_000_000), 1920, 1080, 1, 0, 5)
run("S4 soup 1M 512x512 spp1 (RR depth)", lambda: scenes.random_soup(1_000_000), 512, 512, 1, 0, 5)
run("S5 soup 1M 256x256 spp4 depth 4", lambda: scenes.random_soup(1_000_000), 256, 256, 4, 4, 5)
run("S6 soup 1M 960x540 spp1 depth 8", lambda: scenes.random_soup(1_000_000), 960, 540, 1, 8, 5)
run("S7 soup 1M 960x540 spp1 depth 1", lambda: scenes.random_soup(1_000_000), 960, 540, 1, 1, 5)
run("S8 soup 1M 256x256 spp1 (RR depth)", lambda: scenes.random_soup(1_000_000), 256, 256, 1, 0, 5)
run("S9 Cornell 256x256 spp1 (RR depth)", scenes.cornell_box, 256, 256, 1, 0, 5)
run("S10 soup 1M 384x384 spp1 (RR depth)", lambda: scenes.random_soup(1_000_000), 384, 384, 1, 0, 5)
run("S2 soup 100k 1920x1080 spp4 (RR depth)", lambda: scenes.random_soup(100_000), 1920, 1080, 4, 0, 3)

# glTF configurations: the tree's Duck.gltf does not travel to the GPU box and there is no Sponza-class asset, so both
# use the generated stand-in of scenes.write_rooms_gltf (same loader path); shading is the build-defined
# Lambert over sutil::Scene materials
import pathlib, tempfile
tmp = pathlib.Path(tempfile.mkdtemp())
small = lambda: scenes.from_gltf(scenes.write_rooms_gltf(tmp / "c3.gltf", n_boxes=350))
run("C3-like glTF 4.2k tris 1920x1080 spp64 depth 8, batched", small, 1920, 1080, 64, 8, 1, batch=1)
big = lambda: scenes.from_gltf(scenes.write_rooms_gltf(tmp / "c5.gltf", n_boxes=20_000))
run("C5-like glTF 240k tris 3840x2160 spp4 depth 16, batched (1 GPU)", big, 3840, 2160, 4, 16, 1, batch=1)
run("S3 glTF 240k tris 1920x1080 spp2 (RR depth)", big, 1920, 1080, 2, 0, 2)
