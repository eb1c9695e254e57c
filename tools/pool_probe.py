#!/usr/bin/env python3
"""Schedule counters of k_trace_pool (needs a build with -DMIPT_POOL_STATS): per-wave bulk / drain time in
100 MHz ticks, drain rounds and entries, for rank 0's shard of world 1 and 8."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optix_raytracer_amd import _lib as L, api, pathtracer as PT, scenes
W, H = 1920, 1080
scene = scenes.random_soup(1_000_000)
lines = []
st = PT.make_state(scene, W, H, 1, 0, max_depth=1, log=lambda lv, tag, msg: lines.append((tag, msg)), log_level=4)
stream = torch.cuda.current_stream()
for steal in [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['32'])]:
    for world in (1, 8):
        st.context.set_option(L.OPT_STEAL_MIN, steal)
        ns = api.shard_num_samples(W, H, world)
        compact = torch.zeros((ns, 4), dtype=torch.float32, device=st.device)
        for _ in range(3):
            st.context.launch_shard(st.params, W, H, 0, world, compact.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        del lines[:]
        s = st.context.launch_stats()
        print("steal %d world %d: %s" % (steal, world, s), flush=True)
        for tag, msg in lines:
            if tag == "SCHED":
                print("   ", msg, flush=True)
