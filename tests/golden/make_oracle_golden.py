"""Regenerates tests/golden/oracle_cornell_32x32_spp4.npz from the CPU oracle (regression fixture of the
oracle itself; the reference cannot be run in this environment -- SURVEY.md 8c)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _oracle as O  # noqa: E402
from optix_raytracer_amd import scenes  # noqa: E402

sc = scenes.cornell_box()
s = O.Scene(sc).build_bvh()
accum, frame, st = s.render(scenes.launch_desc(sc, 32, 32, spp=4))
np.savez_compressed(os.path.join(HERE, "oracle_cornell_32x32_spp4.npz"), accum=accum, frame=frame,
                    closest_rays=st["closest_rays"], shadow_rays=st["shadow_rays"])
print(st, accum[..., :3].mean())
