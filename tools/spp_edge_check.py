"""One sample per pixel (every sample draws a work item), two and three (chunks shorter than the pool batch), a sliver of a film and a
thousand samples on few pixels, through the fused kernels against the resident / streaming pipeline: sample counter exact, films equal
bit for bit.  usage (GPU box): python tools/spp_edge_check.py"""
import sys, os, numpy as np
sys.path.insert(0, ".")
from lumillyrender_amd import abi, device, host
bad = 0
for name, other in (("cbox-spheres.toml", abi.LR_FLAG_RESIDENT), ("mesh-box.toml", abi.LR_FLAG_STREAMING), ("ibl-lens.toml", abi.LR_FLAG_STREAMING), ("brdf-row.toml", abi.LR_FLAG_RESIDENT)):
    for (w, h, spp) in ((256, 256, 1), (300, 200, 2), (511, 3, 3), (1024, 1024, 1), (64, 64, 1000)):
        d = host.Description(os.path.join("scenes", name)); d.set_resolution(w, h)
        sc = device.Scene(d)
        a = sc.render(d.render_params(spp=spp, seed=5, flags=abi.LR_FLAG_FUSED)); st = sc.stats()
        b = sc.render(d.render_params(spp=spp, seed=5, flags=other))
        ok = st.samples == w * h * spp and np.array_equal(a, b, equal_nan=True) and st.pipeline == 2
        print(name, w, h, spp, "ok" if ok else "MISMATCH", st.samples, w * h * spp, flush=True)
        bad += not ok
        sc.close()
print("bad:", bad)
