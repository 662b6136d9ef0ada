"""Does the tile edge matter to ONE GPU rendering the whole frame?  (A wave draws 64 consecutive pixel ranks: 64 x 1 pixels of a
64-px tile, 32 x 2 of a 32-px tile, 16 x 4, 8 x 8.)  usage: tile_size_probe.py scene W H spp [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lumillyrender_amd import device, host
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = host.Description(os.path.join(ROOT, "scenes", name)); d.set_resolution(W, H)
sc = device.Scene(d)
film = np.zeros((H, W, 3), dtype=np.float32)
rows = []
full, nf = host.full_tile(W, H)
sc.render(d.render_params(spp=spp, seed=9), full, nf, out=film)
for rnd in range(2):
    for tile in (0, 64, 32, 16, 8):
        tiles, n = (full, nf) if tile == 0 else host.tiles(W, H, tile, 0, 1)
        t0 = time.perf_counter(); sc.render(d.render_params(spp=spp, seed=rnd), tiles, n, out=film); dt = time.perf_counter() - t0
        rows.append({"tile": tile or "whole film", "round": rnd, "tiles": n, "wall_ms": round(dt * 1e3, 3), "device_ms": round(sc.stats().render_ms, 3),
                     "Msamples_s": round(W * H * spp / dt / 1e6, 1)})
        print(json.dumps(rows[-1]), flush=True)
if len(sys.argv) > 5:
    json.dump({"scene": name, "width": W, "height": H, "spp": spp, "rows": rows}, open(sys.argv[5], "w"), indent=1)
