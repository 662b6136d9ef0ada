"""How much of the tree kernel's time is the scene not fitting L2?  Renders mesh-box with the blob at 100k / 36k / 9k triangles
(default pipeline: the fused k_path_tree since round 3; LR_FLAG_STREAMING in argv[1] for the round-2 k_trace)."""
import sys, os, time, tempfile, shutil
sys.path.insert(0, ".")
sys.path.insert(0, "assets")
import gen_assets
from lumillyrender_amd import abi, device, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = int(sys.argv[1]) if len(sys.argv) > 1 else 0
blob = os.path.join(host.ASSET_ROOT, "models/blob/blob.obj")
keep = blob + ".keep"
shutil.copy(blob, keep)
try:
    for n_lon, n_lat in ((250, 201), (150, 121), (75, 61)):
        gen_assets.make_blob(blob, n_lon=n_lon, n_lat=n_lat)
        d = host.Description(os.path.join(ROOT, "scenes", "mesh-box.toml")); d.set_resolution(1920, 1370)
        sc = device.Scene(d)
        tiles, n = host.full_tile(1920, 1370)
        for rep in range(2):
            p = d.render_params(spp=512, seed=rep, flags=abi.LR_FLAG_PROFILE | FLAGS)
            t0 = time.time(); sc.render_device(p, tiles, n); dt = time.time() - t0
        st = device.stats_dict(sc.stats())
        k = st["kernels"]
        per = {n: round(v['ms'] / v['timed'], 3) for n, v in k.items() if v.get('timed')}
        print(f"triangles {2 * n_lon * (n_lat - 1):6d}: {1920 * 1370 * 512 / dt / 1e6:7.1f} Msamples/s  segments/sample {st['segments'] / st['samples']:.2f}  ms/launch {per}", flush=True)
        sc.close()
finally:
    shutil.move(keep, blob)
