"""Ramp and tail of the one-launch kernels: a frame (or rank `r` of `world`'s share of it) through the timeline build
(make -C lumillyrender_amd/csrc timeline -> build/v_timeline.so, loaded through LR_HIP_LIB), which prints when every wave
entered, first saw the work-item dispenser dry, and left.  usage: LR_HIP_LIB=build/v_timeline.so timeline_probe.py scene W H spp [world rank [tile]]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lumillyrender_amd import device, host
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
world = int(sys.argv[5]) if len(sys.argv) > 5 else 1
rank = int(sys.argv[6]) if len(sys.argv) > 6 else 0
tile = int(sys.argv[7]) if len(sys.argv) > 7 else 0
d = host.Description(os.path.join(ROOT, "scenes", name)); d.set_resolution(W, H)
sc = device.Scene(d)
tiles, n = host.tiles(W, H, tile, rank, world)
film = np.zeros((H, W, 3), dtype=np.float32)
for rep in range(3):
    p = d.render_params(spp=spp, seed=rep)
    t0 = time.perf_counter(); sc.render(p, tiles, n, out=film); dt = time.perf_counter() - t0
    print(json.dumps({"scene": name, "spp": spp, "world": world, "rank": rank, "wall_ms": round(dt * 1e3, 3), "device_ms": round(sc.stats().render_ms, 3)}), flush=True)
