"""Multi-GPU plumbing: image-tile shards (StaticWorkDistribution, SDK/sutil/WorkDistribution.h:34-91) +
the one exchange of the path, an all-gather of the compact float4 shards (RCCL on GPUs, gloo in CPU tests).
The reference merges through pinned zero-copy host memory (optixMultiGPU.cu:282-292); here every rank ends
up with the full frame."""
import numpy as np
import torch
import torch.distributed as dist

from . import api


def shard_pixel_indices(width, height, world, rank):
    """Pixel index (y*width+x) of every sample of `rank`; 0xFFFFFFFF for samples outside the image."""
    ns = api.shard_num_samples(width, height, world)
    out = np.full(ns, 0xFFFFFFFF, np.uint32)
    ox = np.arange(32, dtype=np.int64) % 8
    oy = np.arange(32, dtype=np.int64) // 8
    for tile in range(ns // 32):
        px, py = api.shard_pixel(width, height, world, rank, tile * 32)      # tile origin from the C ABI
        x, y = px + ox, py + oy
        ok = (x < width) & (y < height)
        idx = (y * width + x).astype(np.uint32)
        out[tile * 32:(tile + 1) * 32] = np.where(ok, idx, 0xFFFFFFFF)
    return out


def all_gather_shards(compact, world=None):
    """compact: (ns, 4) float32 tensor of this rank -> (world*ns, 4) tensor, rank-major, on every rank."""
    world = world or dist.get_world_size()
    gathered = torch.empty((world * compact.shape[0], 4), dtype=compact.dtype, device=compact.device)
    try:
        dist.all_gather_into_tensor(gathered, compact)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(compact) for _ in range(world)]
        dist.all_gather(parts, compact)
        gathered = torch.cat(parts, 0)
    return gathered


def unpermute_host(gathered, world, width, height):
    """Host-side statement of mipt_unpermute (used by CPU tests): gathered (world*ns,4) -> (h,w,4)."""
    g = np.asarray(gathered, np.float32).reshape(world, -1, 4)
    out = np.zeros((height * width, 4), np.float32)
    for r in range(world):
        idx = shard_pixel_indices(width, height, world, r)
        ok = idx != 0xFFFFFFFF
        out[idx[ok]] = g[r][ok]
    return out.reshape(height, width, 4)
