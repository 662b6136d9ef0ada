"""What one rank of a strong-scaling run does, alone on this GPU: rank 0's tiles of a BASELINE frame at world 1 / 2 / 4 / 8
through lr_render (read-back included).  ideal = t(1) / world; the ratio is the per-rank efficiency the N-GPU line can reach
when the ranks do not disturb each other (fixed costs per call: tile upload, rank table, resolve, read-back, kernel ramp / tail).
usage: strong_rank_probe.py [scene W H spp [tile [out.json]]]"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from lumillyrender_amd import host, device
name = sys.argv[1] if len(sys.argv) > 1 else "cbox-spheres.toml"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1024, 1024, 1024)
tile = int(sys.argv[5]) if len(sys.argv) > 5 else 64
out = sys.argv[6] if len(sys.argv) > 6 else None
d = host.Description("scenes/" + name); d.set_resolution(W, H)
sc = device.Scene(d)
film = np.zeros((H, W, 3), dtype=np.float32)
t1 = None
rows = []
for world in (1, 2, 4, 8):
    tiles, n = host.tiles(W, H, tile, 0, world)
    best, dev_ms = 1e9, 0.0
    for rep in range(4):
        p = d.render_params(spp=spp, seed=rep)
        t0 = time.perf_counter(); sc.render(p, tiles, n, out=film); dt = time.perf_counter() - t0
        if rep and dt < best: best, dev_ms = dt, sc.stats().render_ms
    t1 = t1 or best
    pix = sum(tiles[i].w * tiles[i].h for i in range(n))
    rows.append({"world": world, "tiles": n, "pixels": pix, "wall_ms": round(best * 1e3, 3), "device_ms": round(dev_ms, 3),
                 "ideal_ms": round(t1 / world * 1e3, 3), "efficiency": round(t1 / world / best, 4)})
    print(f"world {world}: rank 0 renders {n} tiles in {best * 1e3:.2f} ms (device {dev_ms:.2f} ms); ideal {t1 / world * 1e3:.2f} ms; efficiency {t1 / world / best:.3f}", flush=True)
if out:
    json.dump({"probe": "one rank's strong-scaling share, alone on one GPU (best of 3 after a warm-up, lr_render incl. read-back)",
               "scene": name, "width": W, "height": H, "spp": spp, "tile": tile, "rows": rows}, open(out, "w"), indent=1)
