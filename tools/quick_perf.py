"""Quick throughput probe on one GPU: python tools/quick_perf.py [scene] [W] [H] [spp] [slots] [flags] [integrator 0=pt 1=pt-direct]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lumillyrender_amd import abi, device, host

name = sys.argv[1] if len(sys.argv) > 1 else "cbox-spheres.toml"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 64
slots = int(sys.argv[5]) if len(sys.argv) > 5 else 0
flags = int(sys.argv[6]) if len(sys.argv) > 6 else abi.LR_FLAG_PROFILE
integ = int(sys.argv[7]) if len(sys.argv) > 7 else None
d = host.Description(os.path.join(ROOT, "scenes", name)); d.set_resolution(W, H)
sc = device.Scene(d)
tiles, n = host.full_tile(W, H)
for rep in range(2):
    p = d.render_params(spp=spp, seed=rep, path_slots=slots, flags=flags, integrator=integ)
    t0 = time.time(); sc.render_device(p, tiles, n); dt = time.time() - t0
    st = device.stats_dict(sc.stats())
    print(json.dumps({"wall_s": round(dt, 4), "Msamples_s": round(W * H * spp / dt / 1e6, 1), **st}))
