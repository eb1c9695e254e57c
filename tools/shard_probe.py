#!/usr/bin/env python3
"""Per-rank cost of the sharded frame on ONE GPU: renders rank 0's shard for world = 1,2,4,8 and prints
per-kernel times (what each GPU of an N-GPU run does, without the all-gather)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optix_raytracer_amd import _lib as L, api, pathtracer as PT, scenes
W, H = 1920, 1080
scene = scenes.random_soup(1_000_000)
st = PT.make_state(scene, W, H, 1, 0, max_depth=1)
stream = torch.cuda.current_stream()
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device=st.device)
# optional arguments: comma lists of MIPT_OPT_STEAL_MIN and MIPT_OPT_BURST values to sweep (default: the library's defaults)
steals = [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else [str(st.context.get_option(L.OPT_STEAL_MIN))])]
drains = [int(x, 0) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else [str(st.context.get_option(L.OPT_BURST))])]
for steal, drain, world in [(s_, d_, w_) for s_ in steals for d_ in drains for w_ in [int(x) for x in os.environ.get('PROBE_WORLDS', '1,2,4,8').split(',')]]:
    st.context.set_option(L.OPT_STEAL_MIN, steal)
    st.context.set_option(L.OPT_BURST, drain)
    ns = api.shard_num_samples(W, H, world)
    compact = torch.zeros((ns, 4), dtype=torch.float32, device=st.device)
    gathered = torch.zeros((world * ns, 4), dtype=torch.float32, device=st.device)
    def step():
        st.context.launch_shard(st.params, W, H, 0, world, compact.data_ptr(), stream=stream.cuda_stream)
        st.context.unpermute(gathered.data_ptr(), world, W, H, st.accum.data_ptr(), frame.data_ptr(), stream=stream.cuda_stream)
    for _ in range(5):
        step()
    st.context.set_option(L.OPT_PROFILE, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    tm = st.context.launch_timings()
    st.context.set_option(L.OPT_PROFILE, 0)
    print("steal %2d burst 0x%04x world %d: rank-0 step %.3f ms (no all-gather) | closest %.3f any %.3f shade %.3f other %.3f" % (
        steal, drain, world, dt, tm["ms_trace_closest"] / K, tm["ms_trace_any"] / K, tm["ms_shade"] / K, (tm["ms_raygen"] + tm["ms_resolve"]) / K), flush=True)
