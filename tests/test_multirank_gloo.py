"""world_size-2 (and 3) run of the multi-GPU driver logic on CPU over gloo.  The GPU render call is
replaced by the oracle here (tests may use it); what is under test is the tile sharding, the zeroed
per-rank films and the host gather: the assembled film must equal a single-rank render bit for bit."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import scene_path

W, H, SPP, TILE = 40, 28, 4, 16


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_path, packed=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lumillyrender_amd import host, multigpu
    from oracle import binding as oracle
    desc = host.Description(scene_path("cbox-spheres.toml"))
    desc.set_resolution(W, H)
    params = desc.render_params(spp=SPP, seed=4)

    def render_fn(tiles, n, out):
        img = oracle.render(desc, params, tiles, n, threads=1)
        out += img                                   # oracle leaves untouched pixels at zero
    film = multigpu.render_sharded(render_fn, W, H, TILE, rank, world)
    mine = (film != 0).any(axis=2).sum()
    counts = torch.tensor([int(mine)])
    dist.all_reduce(counts)
    if packed:
        multigpu.gather_tiles(film, W, H, TILE, dist, dst=0)
    else:
        multigpu.gather_film(film, dist, dst=0)
    if rank == 0:
        np.save(out_path, film)
        assert int(counts.item()) <= W * H
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,packed", [(2, False), (3, False), (2, True), (3, True)])
def test_sharded_render_equals_single_rank(tmp_path, world, packed):
    out_path = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(world, _free_port(), out_path, packed), nprocs=world, join=True)
    from lumillyrender_amd import host
    from oracle import binding as oracle
    desc = host.Description(scene_path("cbox-spheres.toml"))
    desc.set_resolution(W, H)
    ref = oracle.render(desc, desc.render_params(spp=SPP, seed=4), threads=2)
    got = np.load(out_path)
    assert np.array_equal(got, ref)


def _shared_worker(rank, world, port, out_path, W=W, H=H):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lumillyrender_amd import host, multigpu
    from oracle import binding as oracle
    desc = host.Description(scene_path("cbox-spheres.toml"))
    desc.set_resolution(W, H)
    params = desc.render_params(spp=SPP, seed=4)
    film = multigpu.SharedFilm(W, H, TILE, dist, dst=0)
    assert film.array.shape == (H, W, 3)
    assert film.shared
    assert film.name and not os.path.exists(film.name)       # unlinked once every rank had mapped it: a crash leaks nothing
    tiles, n = multigpu.shard_tiles(W, H, TILE, rank, world)
    frames = []
    for frame in range(2):                           # two frames through one film: collect / release keep them apart
        p = desc.render_params(spp=SPP, seed=4 + frame)
        img = oracle.render(desc, p, tiles, n, threads=1)
        for i in range(n):                           # like lr_render: only the pixels of the tiles given
            t = tiles[i]
            film.array[t.y0:t.y0 + t.h, t.x0:t.x0 + t.w] = img[t.y0:t.y0 + t.h, t.x0:t.x0 + t.w]
        got = film.collect()
        if rank == 0:
            frames.append(np.array(got))             # dst consumes the frame ...
        film.release()                               # ... before anybody renders the next one into the film
    if rank == 0:
        np.save(out_path, np.stack(frames))
    film.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,film", [(2, (W, H)), (3, (W, H)), (8, (W, H)), (2, (37, 29))])
def test_shared_memory_film(tmp_path, world, film):
    """bench.py's single-node path: every rank writes its tiles into one film in /dev/shm.  world = 8 is the node the
    driver's scaling run uses (here the 40 x 28 film has six 16-pixel tiles: two of the eight ranks hold NO tile).  The 37 x 29
    film has a float count that is not a multiple of 16: the barrier words sit behind a padded film (the film view
    took the padded length and could not be reshaped)."""
    W, H = film
    out_path = str(tmp_path / "film.npy")
    mp.spawn(_shared_worker, args=(world, _free_port(), out_path, W, H), nprocs=world, join=True)
    from lumillyrender_amd import host
    from oracle import binding as oracle
    desc = host.Description(scene_path("cbox-spheres.toml"))
    desc.set_resolution(W, H)
    got = np.load(out_path)
    for frame in range(2):
        ref = oracle.render(desc, desc.render_params(spp=SPP, seed=4 + frame), threads=2)
        assert np.array_equal(got[frame], ref), frame


def _killed_worker(rank, world, port, victim):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lumillyrender_amd import multigpu
    film = multigpu.SharedFilm(W, H, TILE, dist, dst=0)
    assert film.shared and not os.path.exists(film.name)
    film.array[rank % H, 0, 0] = 1.0
    if rank == victim:
        os.kill(os.getpid(), 9)                      # a rank dies mid-frame (driver kill, OOM): no close(), no barrier
    film.collect()
    film.close()
    dist.destroy_process_group()


def test_a_killed_rank_leaves_nothing_in_dev_shm():
    """The film's backing file loses its name as soon as every rank has mapped it, so eight ranks of which one is killed
    between two barriers leave no /dev/shm segment behind (the pages go with the last mapping)."""
    import glob
    before = set(glob.glob("/dev/shm/lumilly_film_*"))
    with pytest.raises(Exception):
        mp.spawn(_killed_worker, args=(8, _free_port(), 5), nprocs=8, join=True)
    assert set(glob.glob("/dev/shm/lumilly_film_*")) <= before


def _subgroup_worker(rank, world, port, out_path):
    """gather on a SUB-group whose dst is not global rank 0: `dst` is a rank of the group."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lumillyrender_amd import host, multigpu
    from oracle import binding as oracle
    grp = dist.new_group(ranks=[1, 2])
    if rank in (1, 2):
        desc = host.Description(scene_path("cbox-spheres.toml"))
        desc.set_resolution(W, H)
        params = desc.render_params(spp=SPP, seed=4)
        g_rank, g_world = dist.get_rank(grp), dist.get_world_size(grp)
        films = []
        for fn in (multigpu.gather_film, None):
            tiles, n = multigpu.shard_tiles(W, H, TILE, g_rank, g_world)
            film = oracle.render(desc, params, tiles, n, threads=1)
            if fn is not None:
                fn(film, dist, dst=0, group=grp)
            else:
                multigpu.gather_tiles(film, W, H, TILE, dist, dst=0, group=grp)
            films.append(film)
        if g_rank == 0:
            assert rank == 1
            np.save(out_path, np.stack(films))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_on_a_subgroup(tmp_path):
    out_path = str(tmp_path / "film.npy")
    mp.spawn(_subgroup_worker, args=(3, _free_port(), out_path), nprocs=3, join=True)
    from lumillyrender_amd import host
    from oracle import binding as oracle
    desc = host.Description(scene_path("cbox-spheres.toml"))
    desc.set_resolution(W, H)
    ref = oracle.render(desc, desc.render_params(spp=SPP, seed=4), threads=2)
    got = np.load(out_path)
    assert np.array_equal(got[0], ref) and np.array_equal(got[1], ref)


def _shm_barrier_worker(rank, world, path, rounds):
    from lumillyrender_amd import host
    m = np.memmap(path, dtype=np.uint32, mode="r+", shape=(64,))
    state = m[48:].ctypes.data                                      # two words on a cache line of their own
    for it in range(1, rounds + 1):
        m[rank] = it                                                 # plain store before arriving ...
        host.shm_barrier(state, world, 30.0)
        assert all(int(m[r]) >= it for r in range(world)), (rank, it, [int(m[r]) for r in range(world)])   # ... is visible to everyone who leaves
        host.shm_barrier(state, world, 30.0)                         # (nobody overwrites its slot before everyone has looked)


def test_shared_memory_barrier(tmp_path):
    """lr_host_shm_barrier (the per-frame barrier of multigpu.SharedFilm on one node): 4 processes, 2000 rounds, every store made
    before a rank arrives is seen by every rank that leaves; a barrier the others never reach returns an error after its timeout
    instead of spinning for ever."""
    from lumillyrender_amd import host
    path = tmp_path / "barrier.u32"
    np.zeros(64, dtype=np.uint32).tofile(path)
    mp.spawn(_shm_barrier_worker, args=(4, str(path), 2000), nprocs=4, join=True)
    m = np.memmap(path, dtype=np.uint32, mode="r+", shape=(64,))
    assert int(m[48]) == 0 and int(m[49]) == 4000                   # count back at zero, 2 x 2000 rounds
    t0 = time.time()
    with pytest.raises(host.LumillyError):
        host.shm_barrier(m[48:].ctypes.data, 2, 0.2)                 # world 2, one arrival
    assert time.time() - t0 < 5.0
    # ... and is BROKEN from then on: a late rank must not complete the round alone (every later call fails at once)
    assert int(m[50]) == 1
    t0 = time.time()
    with pytest.raises(host.LumillyError):
        host.shm_barrier(m[48:].ctypes.data, 2, 30.0)
    assert time.time() - t0 < 1.0
    host.shm_barrier(m[48:].ctypes.data, 1, 0.2)                     # world 1: nothing to wait for
