"""Resident vs streaming pipeline on random tree scenes of growing size.  usage: pipeline_probe.py [W H spp]
Prints Msamples/s of both pipelines (forced by flag) and which one lr_render picks by itself."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import importlib.util
spec = importlib.util.spec_from_file_location("fz", "tools/fuzz_parity.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from lumillyrender_amd import abi, device, host

W = int(sys.argv[1]) if len(sys.argv) > 1 else 768
H = int(sys.argv[2]) if len(sys.argv) > 2 else 768
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 256
for seed, max_objs in ((11, 40), (12, 80), (13, 150), (14, 300), (15, 600), (16, 1500), (17, 4000)):
    text, integ, cam = fz.scene_text(seed, W, H, max_objs)
    desc = host.Description(text=text)
    sc = device.Scene(desc)
    tiles, n = host.full_tile(W, H)
    out = {}
    for name, flag in (("auto", 0), ("resident", abi.LR_FLAG_RESIDENT), ("streaming", abi.LR_FLAG_STREAMING)):
        best = 0.0
        for rep in range(2):
            p = desc.render_params(spp=spp, seed=rep, flags=flag)
            t0 = time.time(); sc.render_device(p, tiles, n); dt = time.time() - t0
            best = max(best, W * H * spp / dt / 1e6)
        st = device.stats_dict(sc.stats())
        out[name] = (round(best, 1), st["pipeline"], st["path_slots"])
    print(f"prims {desc.desc.n_prims:5d} {integ:9s}: auto {out['auto']}  resident {out['resident']}  streaming {out['streaming']}")
    sc.close()
