"""One process per GPU: shard the pixel tile queue over ranks, render, gather the film on rank 0.

The path shards without any exchange step (pixels are independent jobs in the reference,
main.rs:73-126): every rank holds the whole scene, renders the tiles lr_host_tiles deals to it (tile (i, j) of the 16-px
tile grid -> rank (i + k j) mod world, k coprime with world: no stripes whatever the grid width), and the film is assembled on the host -- the counterpart of the reference's channel drain
(main.rs:129-132): on one node every rank's lr_render writes its tiles straight into ONE film in POSIX shared
memory (SharedFilm; lr_render only touches the pixels of the tiles it is given) and a barrier publishes it; across
nodes rank 0 gathers each rank's packed tiles over gloo (gather_tiles).  No RCCL collective is on the data path.  RNG keys are (seed, pixel, sample), so the assembled film is bit-identical to a
single-rank render.
"""
import os
import time

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
    # `dst` is a rank of `group`; torch.distributed.reduce wants the global rank
    dist.reduce(t, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
    return film


class _GatherPlan:
    """Tile rectangles of every rank and reusable send / receive buffers for one (film size, tile, world, rank)."""

    def __init__(self, width, height, tile, world, rank, is_dst):
        import torch
        self.rects = []                                            # per rank: [(y0, y1, x0, x1, offset_in_floats), ...]
        sizes = []
        for r in range(world):
            tl, n = shard_tiles(width, height, tile, r, world)
            off, rr = 0, []
            for i in range(n):
                t = tl[i]
                rr.append((t.y0, t.y0 + t.h, t.x0, t.x0 + t.w, off))
                off += t.w * t.h * 3
            self.rects.append(rr)
            sizes.append(off)
        cap = max(max(sizes), 1)
        self.send = np.zeros(cap, dtype=np.float32)
        self.send_t = torch.from_numpy(self.send)                  # shares memory with self.send
        self.recv = [np.zeros(cap, dtype=np.float32) for _ in range(world)] if is_dst else None
        self.recv_t = [torch.from_numpy(a) for a in self.recv] if is_dst else None


_plans = {}


def gather_tiles(film, width, height, tile, dist=None, dst=0, group=None):
    """Collect the tile pixels every rank rendered into `film` on `dst`: each rank packs ITS tiles (1/world of the
    film) and one gloo gather moves them, instead of reducing `world` whole films.  Buffers are allocated once per
    film geometry and reused.  Returns the film on dst."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return film
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    key = (width, height, tile, world, rank, dst)
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = _GatherPlan(width, height, tile, world, rank, rank == dst)
    for y0, y1, x0, x1, off in plan.rects[rank]:
        plan.send[off:off + (y1 - y0) * (x1 - x0) * 3] = film[y0:y1, x0:x1].reshape(-1)
    dist.gather(plan.send_t, plan.recv_t, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
    if rank == dst:
        for r in range(world):
            if r == rank:
                continue
            flat = plan.recv[r]
            for y0, y1, x0, x1, off in plan.rects[r]:
                film[y0:y1, x0:x1] = flat[off:off + (y1 - y0) * (x1 - x0) * 3].reshape(y1 - y0, x1 - x0, 3)
    return film


_BARRIER_WORDS = 16          # floats appended to the shared film file: 64 bytes, of which lr_host_shm_barrier uses three uint32 words (arrivals, round, broken)


class SharedFilm:
    """The film all ranks of one node render into.  `array` is an (H, W, 3) float32 view of a file in /dev/shm mapped by
    every rank; `collect()` makes the ranks' tiles visible on rank `dst` (a barrier).  Ranks on different hosts (or no
    process group) get private arrays and `collect()` falls back to gather_tiles.

    Frame loop protocol: render -> collect() -> [dst reads the frame] -> release() -> next render.  `release()` is the
    second barrier that keeps the other ranks from rendering frame k+1 into the film while dst still reads frame k; a
    loop whose consumer only looks at the last frame (bench.py) may skip it.  The backing file is unlinked as soon as
    every rank has mapped it, so a crashed job leaves nothing in /dev/shm."""

    def __init__(self, width, height, tile, dist=None, dst=0, group=None, timeout_s=1800.0):
        import socket
        self.width, self.height, self.tile, self.dist, self.dst, self.group = width, height, tile, dist, dst, group
        self.timeout_s = float(timeout_s)   # of one barrier (a frame, or dst's consume step between collect() and release()); gloo's own default is 30 min
        self._state = None
        self.path = None                 # set only while the backing file still has a name (construction)
        self.name = None                 # the name it had, for diagnostics
        self.shared = False
        active = dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1
        if active:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            hosts = [None] * world
            dist.all_gather_object(hosts, socket.gethostname(), group=group)
            self.shared = len(set(hosts)) == 1 and os.path.isdir("/dev/shm")
        if self.shared:
            n_pix3 = height * width * 3
            n_film = n_pix3 + (-n_pix3) % 16                          # the barrier words start on a 64-byte boundary behind the (padded) film
            box = [None]
            if rank == dst:
                self.path = f"/dev/shm/lumilly_film_{os.getpid()}_{time.time_ns()}.f32"
                self._map = np.memmap(self.path, dtype=np.float32, mode="w+", shape=(n_film + _BARRIER_WORDS,))     # zero-filled: film + barrier state
                box[0] = self.path
            src = dist.get_global_rank(group, dst) if group is not None else dst
            dist.broadcast_object_list(box, src=src, group=group)
            if rank != dst:
                self.path = box[0]
                self._map = np.memmap(self.path, dtype=np.float32, mode="r+", shape=(n_film + _BARRIER_WORDS,))
            self.array = self._map[:n_pix3].reshape(height, width, 3)
            self._state = self._map[n_film:].ctypes.data             # two uint32 words on a cache line of their own (lr_host_shm_barrier)
            self._world = world
            self.name = self.path
            self._owner = rank == dst
            dist.barrier(group=group)                                 # every rank holds its mapping ...
            if self._owner:
                os.unlink(self.path)                                  # ... so the name can go: the pages live as long as the mappings
            self.path = None
            dist.barrier(group=group)                                 # nobody returns while the name still exists
        else:
            self.array = np.zeros((height, width, 3), dtype=np.float32)
            self._owner = False

    def _barrier(self):
        # the ranks of one node meet on two words of the shared film file itself (lr_host_shm_barrier): a gloo barrier of 8 processes
        # takes 0.3-0.6 ms, 2-3 % of a rank's 22-ms share of the headline frame; this one takes microseconds
        if self._state is None:
            raise RuntimeError("SharedFilm is closed")
        host.shm_barrier(self._state, self._world, self.timeout_s)

    def collect(self):
        """After every rank rendered its tiles into `array`: the complete film is readable on dst."""
        if self.shared:
            self._barrier()
        else:
            gather_tiles(self.array, self.width, self.height, self.tile, self.dist, dst=self.dst, group=self.group)
        return self.array

    def release(self):
        """dst has consumed the frame: the ranks may render the next one into the film."""
        if self.shared:
            self._barrier()

    def close(self):
        if self.shared:
            self.dist.barrier(group=self.group)
            self._state = None                                      # (points into the mapping that goes away below)
            arr, self.array, self._map = self.array, None, None
            del arr
            self.shared = False
