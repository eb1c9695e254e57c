"""N>1 path on CPU: world_size-2 and -3 gloo jobs (torch.distributed.run, rendezvous on 127.0.0.1)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,w,h", [(2, 64, 24), (3, 50, 20)])
def test_sharded_frame_reassembles_over_gloo(world, w, h):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world),
           os.path.join(HERE, "_dist_worker.py"), str(w), str(h), "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_sharded_frame_through_the_library_on_ranks_sharing_the_gpu():
    """world 2, gloo, both ranks on the box's one GPU: every rank renders its shard with mipt_launch_shard and assembles the
    frame with mipt_unpermute (tests/_dist_worker.py takes the GPU path when a GPU is present)."""
    import torch
    assert torch.cuda.is_available()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(HERE, "_dist_worker.py"), "200", "96", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
