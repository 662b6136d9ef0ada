"""What EVERY rank of a strong-scaling run does, each alone on this GPU: the tiles lr_host_tiles deals to rank r of `world`
(r = 0 .. world-1) of a BASELINE frame, through lr_render (read-back included), for world 1 / 2 / 4 / 8.  The N-GPU frame ends
with its slowest rank, so the figure of merit is  predicted_efficiency = t(1) / world / max_r t(r)  -- not rank 0's time (round 4
timed rank 0 only, which on a column-striped split is the lightest rank).  Also records each rank's work counters
(segments + shadow rays), the proxy the judge used.  No inter-rank effect is in this number: it is the tile split plus every
per-rank fixed cost (tile upload, rank table, kernel ramp and tail, resolve, read-back).
usage: strong_rank_probe.py scene W H spp [tile [out.json [reps]]]"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from lumillyrender_amd import host, device
name = sys.argv[1] if len(sys.argv) > 1 else "cbox-spheres.toml"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1024, 1024, 1024)
tile = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # 0: the library's default (lr_host_default_tile)
out = sys.argv[6] if len(sys.argv) > 6 else None
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 2
if tile <= 0:
    tile = host.default_tile() if hasattr(host, "default_tile") else 64
d = host.Description("scenes/" + name); d.set_resolution(W, H)
sc = device.Scene(d)
film = np.zeros((H, W, 3), dtype=np.float32)
p = d.render_params(spp=spp, seed=0)
full, nfull = host.tiles(W, H, tile, 0, 1)
sc.render(p, full, nfull, out=film)                          # warm-up at full size: buffers, clocks
t1 = None
rows = []
for world in (1, 2, 4, 8):
    ranks = []
    for r in range(world):
        tiles, n = host.tiles(W, H, tile, r, world)
        best, dev_ms, st = 1e9, 0.0, None
        for rep in range(reps):
            t0 = time.perf_counter(); sc.render(p, tiles, n, out=film); dt = time.perf_counter() - t0
            if dt < best:
                best, s = dt, sc.stats()
                dev_ms, st = s.render_ms, (int(s.segments), int(s.shadow_rays))
        pix = sum(tiles[i].w * tiles[i].h for i in range(n))
        ranks.append({"rank": r, "tiles": n, "pixels": pix, "wall_ms": round(best * 1e3, 3), "device_ms": round(dev_ms, 3),
                      "segments": st[0], "shadow_rays": st[1]})
    wall = [q["wall_ms"] for q in ranks]
    work = [q["segments"] + q["shadow_rays"] for q in ranks]
    t1 = t1 or wall[0]
    row = {"world": world, "max_ms": max(wall), "mean_ms": round(float(np.mean(wall)), 3), "min_ms": min(wall),
           "ideal_ms": round(t1 / world, 3), "predicted_efficiency": round(t1 / world / max(wall), 4),
           "rank0_efficiency": round(t1 / world / wall[0], 4),
           "work_max_over_mean": round(max(work) / float(np.mean(work)), 4), "work_min_over_mean": round(min(work) / float(np.mean(work)), 4),
           "time_max_over_mean": round(max(wall) / float(np.mean(wall)), 4), "ranks": ranks}
    rows.append(row)
    print(f"world {world}: max {row['max_ms']:.2f} mean {row['mean_ms']:.2f} min {row['min_ms']:.2f} ms; ideal {row['ideal_ms']:.2f}; "
          f"predicted efficiency {row['predicted_efficiency']:.3f} (rank 0 alone: {row['rank0_efficiency']:.3f}); "
          f"work max/mean {row['work_max_over_mean']:.3f}", flush=True)
if out:
    json.dump({"probe": "every rank's strong-scaling share, each alone on one GPU (best of %d, lr_render incl. read-back); "
                        "predicted_efficiency = t(1) / world / max over ranks" % reps,
               "scene": name, "width": W, "height": H, "spp": spp, "tile": tile,
               "split": host.tile_split_name() if hasattr(host, "tile_split_name") else "tile id % world on 64-px tiles (round 4)",
               "rows": rows}, open(out, "w"), indent=1)
