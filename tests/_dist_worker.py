"""Worker of tests/test_dist_cpu.py: one rank of a world_size-N gloo job.  Without a GPU each rank renders its image-tile
shard with the CPU oracle (standing in for the GPU it does not have); WITH a GPU (`-m gpu` test: the ranks share the box's
one card) each rank renders its shard with mipt_launch_shard and scatters the gathered shards with mipt_unpermute -- the
product path of bench.py at N > 1 -- and the oracle only checks.  The shards are all-gathered with the SAME helper bench.py
uses, and every rank checks the reassembled frame."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402
from optix_raytracer_amd import dist as D  # noqa: E402
from optix_raytracer_amd import scenes  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w, h, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    scene = scenes.cornell_box()
    osc = O.Scene(scene).build_bvh()
    launch = scenes.launch_desc(scene, w, h, spp)
    idx = D.shard_pixel_indices(w, h, world, rank)
    if torch.cuda.is_available():
        from optix_raytracer_amd import api, pathtracer as PT
        pst = PT.make_state(scene, w, h, spp, 0, max_depth=2)
        ns = api.shard_num_samples(w, h, world)
        assert ns == idx.shape[0]
        d_compact = torch.zeros((ns, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        pst.context.launch_shard(pst.params, w, h, rank, world, d_compact.data_ptr(), stream=pst.stream.cuda_stream)
        pst.context.sync()
        stats = pst.context.launch_stats()
        st = {"closest_rays": stats["closest_rays"], "shadow_rays": stats["shadow_rays"]}
        ref_compact, ref_st = osc.render_pixels(launch, idx, max_depth=2)          # the oracle is the checker here
        valid = idx != 0xFFFFFFFF
        assert np.array_equal(d_compact.cpu().numpy()[valid].view(np.uint32), ref_compact[valid].view(np.uint32)), "rank %d: shard differs from the oracle" % rank
        assert st["closest_rays"] == ref_st["closest_rays"] and st["shadow_rays"] == ref_st["shadow_rays"]
        gathered = D.all_gather_shards(d_compact.cpu(), world).cuda()            # gloo: staged through the host
        d_acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        d_frame = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        pst.context.unpermute(gathered.data_ptr(), world, w, h, d_acc.data_ptr(), d_frame.data_ptr(), stream=pst.stream.cuda_stream)
        pst.context.sync()
        frame = d_acc.cpu().numpy()
        PT.cleanupState(pst)
    else:
        compact, st = osc.render_pixels(launch, idx, max_depth=2)
        gathered = D.all_gather_shards(torch.from_numpy(compact), world)
        frame = D.unpermute_host(gathered.numpy(), world, w, h)
    full, _, _ = osc.render(launch, max_depth=2)
    ok = np.array_equal(frame.view(np.uint32), full.view(np.uint32))
    # rays traced by all ranks add up to the single-process count
    rays = torch.tensor([st["closest_rays"], st["shadow_rays"]], dtype=torch.int64)
    dist.all_reduce(rays)
    _, _, st_full = osc.render(launch, max_depth=2)
    ok = ok and int(rays[0]) == st_full["closest_rays"] and int(rays[1]) == st_full["shadow_rays"]
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 3)


if __name__ == "__main__":
    main()
