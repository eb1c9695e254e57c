import sys
sys.path.insert(0, "/root/repo")
import torch
from optix_raytracer_amd import _lib as L, pathtracer as PT, scenes
scene = scenes.random_soup(1_000_000)
st = PT.make_state(scene, 1920, 1080, 1, 0, max_depth=1)
ob = PT.OutputBuffer(1920, 1080, st.device)
st.context.set_option(L.OPT_COUNT_VISITS, 1)
PT.launchSubframe(ob, st)
s = st.context.launch_stats()
print(s)
c = s["closest_rays"]; sh = s["shadow_rays"]
print("closest: %.1f nodes + %.1f tris per ray" % ((s["node_visits"] - s["shadow_node_visits"]) / c, (s["tri_tests"] - s["shadow_tri_tests"]) / c))
print("shadow : %.1f nodes + %.1f tris per ray" % (s["shadow_node_visits"] / sh, s["shadow_tri_tests"] / sh))
