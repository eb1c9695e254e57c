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
