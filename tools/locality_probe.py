"""Is a rank's 1/8 share slower per pixel because its pixels are spread over the whole film (L2 locality of the rays in flight)?
Renders the same NUMBER of pixels of a frame (a) as rank 3's diagonal share, (b) as one horizontal strip, (c) as the same strip
in four sequential sub-strips, (d) as 1/8 of the whole-frame time.  usage: locality_probe.py scene W H spp [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lumillyrender_amd import abi, device, host
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = host.Description(os.path.join(ROOT, "scenes", name)); d.set_resolution(W, H)
sc = device.Scene(d)
film = np.zeros((H, W, 3), dtype=np.float32)
p = d.render_params(spp=spp, seed=1)
def strip(y0, h, tile=16):
    lst = [(x, y, min(tile, W - x), min(tile, y0 + h - y)) for y in range(y0, y0 + h, tile) for x in range(0, W, tile)]
    arr = (abi.LrTile * len(lst))()
    for q, (x, y, w, hh) in enumerate(lst): arr[q].x0, arr[q].y0, arr[q].w, arr[q].h = x, y, w, hh
    return arr, len(lst)
def timed(tiles, n, reps=2):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); sc.render(p, tiles, n, out=film); best = min(best, time.perf_counter() - t0)
    return best * 1e3
full, nf = host.tiles(W, H, 0, 0, 1)
t_full = timed(full, nf)
share, ns = host.tiles(W, H, 0, 3, 8)
t_share = timed(share, ns)
hs = H // 8
rows = {"whole_frame_ms": round(t_full, 2), "ideal_eighth_ms": round(t_full / 8, 2), "diagonal_share_ms": round(t_share, 2)}
for y0 in (0, 3 * hs, 7 * hs):
    a, n = strip(y0, hs); rows[f"strip_y{y0}_ms"] = round(timed(a, n), 2)
allstrips = sum(timed(*strip(k * hs, hs), reps=1) for k in range(8))
rows["eight_strips_sum_ms"] = round(allstrips, 2)
q = hs // 4
rows["strip_y%d_in_4_substrips_ms" % (3 * hs)] = round(sum(timed(*strip(3 * hs + k * q, q), reps=1) for k in range(4)), 2)
print(json.dumps(rows))
if len(sys.argv) > 5: json.dump(dict(scene=name, width=W, height=H, spp=spp, **rows), open(sys.argv[5], "w"), indent=1)
