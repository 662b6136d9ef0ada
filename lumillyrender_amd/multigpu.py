"""One process per GPU: shard the pixel tile queue over ranks, render, gather the film on rank 0.

The path shards without any exchange step (pixels are independent jobs in the reference,
main.rs:73-126): every rank holds the whole scene, renders the tiles `i % world == rank` of the
row-major tile grid into a zeroed film, and rank 0 collects every rank's tile pixels over the host
(gloo gather of the packed tiles, 1/world of the film per rank) -- the counterpart of the reference's
channel drain (main.rs:129-132).  No RCCL collective is on the data path.  RNG keys are (seed, pixel, sample), so the assembled film is bit-identical to a
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


def _pack(film, tiles, n):
    if n == 0:
        return np.zeros(0, dtype=np.float32)
    return np.concatenate([film[t.y0:t.y0 + t.h, t.x0:t.x0 + t.w].reshape(-1) for t in (tiles[i] for i in range(n))])


def gather_tiles(film, width, height, tile, dist=None, dst=0, group=None):
    """Collect the tile pixels every rank rendered into `film` on `dst`: each rank packs ITS tiles (1/world of the
    film) and one gloo gather moves them, instead of reducing `world` whole films.  Returns the film on dst."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return film
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shards = [shard_tiles(width, height, tile, r, world) for r in range(world)]
    sizes = [sum(tl[i].w * tl[i].h for i in range(n)) * 3 for tl, n in shards]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.float32)
    mine = _pack(film, *shards[rank])
    buf[:mine.size] = torch.from_numpy(mine)
    recv = [torch.empty(cap, dtype=torch.float32) for _ in range(world)] if rank == dst else None
    dist.gather(buf, recv, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
    if rank == dst:
        for r, (tl, n) in enumerate(shards):
            if r == rank:
                continue
            flat, off = recv[r].numpy(), 0
            for i in range(n):
                t = tl[i]
                film[t.y0:t.y0 + t.h, t.x0:t.x0 + t.w] = flat[off:off + t.w * t.h * 3].reshape(t.h, t.w, 3)
                off += t.w * t.h * 3
    return film
