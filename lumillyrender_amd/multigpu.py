"""One process per GPU: shard the pixel tile queue over ranks, render, gather the film on rank 0.

The path shards without any exchange step (pixels are independent jobs in the reference,
main.rs:73-126): every rank holds the whole scene, renders the tiles `i % world == rank` of the
row-major tile grid into a zeroed film, and the disjoint films are summed onto rank 0 over the host
(gloo) -- the counterpart of the reference's channel drain (main.rs:129-132).  No RCCL collective is on
the data path.  RNG keys are (seed, pixel, sample), so the assembled film is bit-identical to a
single-rank render.
"""
import numpy as np

from . import host


def shard_tiles(width, height, tile, rank, world):
    return host.tiles(width, height, tile, rank, world)


def render_sharded(render_fn, width, height, tile, rank, world, out=None):
    """render_fn(tiles, n_tiles, out) renders the given tiles into `out` (H, W, 3) f32 in place."""
    if out is None:
        out = np.zeros((height, width, 3), dtype=np.float32)
    tiles, n = shard_tiles(width, height, tile, rank, world)
    render_fn(tiles, n, out)
    return out


def gather_film(film, dist=None, dst=0, group=None):
    """Sum the disjoint per-rank films onto `dst` (host tensors; `group` = a gloo group when the default
    group is NCCL).  Returns the film on dst."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return film
    import torch
    t = torch.from_numpy(film)
    dist.reduce(t, dst=dst, group=group)
    return film
