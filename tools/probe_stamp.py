import os, sys
sys.path.insert(0, os.getcwd())
import torch
from optix_raytracer_amd import _lib as L, scenes, pathtracer as PT
scene = scenes.random_soup(1_000_000)
st = PT.make_state(scene, 1920, 1080, 1, 0, max_depth=1)
frame = torch.zeros((1080, 1920, 4), dtype=torch.uint8, device=st.device)
st.params.frame_buffer = frame.data_ptr()
for _ in range(3):
    st.context.launch(st.params, 1920, 1080, 1)
os.environ["MIPT_PROBE_PRINT"] = "1"
print(st.context.launch_stats())
