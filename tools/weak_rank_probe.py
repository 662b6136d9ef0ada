"""One rank's share of the weak-scaling bench at world N, run alone on this GPU: tiles i % N == 0 of the 1024^2 frame at
1024 * N spp (the same number of samples as the N = 1 workload).  usage: weak_rank_probe.py N"""
import sys, time
sys.path.insert(0, ".")
from lumillyrender_amd import host, device
N = int(sys.argv[1])
d = host.Description("scenes/cbox-spheres.toml"); d.set_resolution(1024, 1024)
sc = device.Scene(d)
tiles, n = host.tiles(1024, 1024, 64, 0, N)
for rep in range(3):
    p = d.render_params(spp=1024 * N, seed=rep, integrator=1)
    t0 = time.time(); sc.render_device(p, tiles, n); dt = time.time() - t0
pix = sum(tiles[i].w * tiles[i].h for i in range(n))
print(f"world {N}: {pix} pixels x {1024 * N} spp in {dt * 1e3:.1f} ms = {pix * 1024 * N / dt / 1e6:.0f} Msamples/s per GPU")
