"""What one rank of a strong-scaling run does, alone on this GPU: rank 0's tiles of the configs[1] frame at world 1 / 2 / 4 / 8
through lr_render (read-back included).  ideal = t(1) / world; the ratio is the per-rank efficiency the 8-GPU line can reach
when the ranks do not disturb each other (fixed costs per call: tile upload, rank table, resolve, read-back).
usage: strong_rank_probe.py [scene W H spp]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from lumillyrender_amd import host, device
name = sys.argv[1] if len(sys.argv) > 1 else "cbox-spheres.toml"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1024, 1024, 1024)
d = host.Description("scenes/" + name); d.set_resolution(W, H)
sc = device.Scene(d)
film = np.zeros((H, W, 3), dtype=np.float32)
t1 = None
for world in (1, 2, 4, 8):
    tiles, n = host.tiles(W, H, 64, 0, world)
    best = 1e9
    for rep in range(4):
        p = d.render_params(spp=spp, seed=rep)
        t0 = time.perf_counter(); sc.render(p, tiles, n, out=film); dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    st = sc.stats()
    t1 = t1 or best
    print(f"world {world}: rank 0 renders {n} tiles in {best * 1e3:.2f} ms (device {st.render_ms:.2f} ms); ideal {t1 / world * 1e3:.2f} ms; efficiency {t1 / world / best:.3f}")
