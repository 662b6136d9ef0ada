"""Where the time of config 5 goes: the scene with one feature replaced at a time (uniform sky for the IBL map, pinhole for the thin
lens, Lambert for the GGX mesh, pt for pt-direct), all through the default (fused) pipeline.  usage: c5_breakdown.py [W H spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lumillyrender_amd import device, host
W, H, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 2048, 128)
text = open(os.path.join(ROOT, "scenes", "ibl-lens.toml")).read()
variants = {
    "as stated": lambda t: t,
    "uniform sky": lambda t: t.replace('type = "ibl"\npath = "models/ibl/sky_3k.hdr"', 'type = "uniform"\ncolor = [1, 1, 1]'),
    "pinhole": lambda t: t.replace('type = "thin-lens"', 'type = "ideal-pinhole"').replace('focus-distance = 1800\nf-number = 1.8\n', ''),
    "lambert mesh": lambda t: t.replace('mesh = "blob"\nmaterial = "gloss"', 'mesh = "blob"\nmaterial = "matte"'),
    "pt": lambda t: t.replace('integrator = "pt-direct"', 'integrator = "pt"'),
}
for name, edit in variants.items():
    t = edit(text)
    assert name == "as stated" or t != text, name
    d = host.Description(text=t)
    d.set_resolution(W, H)
    sc = device.Scene(d)
    tiles, n = host.full_tile(W, H)
    best = 1e9
    for rep in range(3):
        p = d.render_params(spp=spp, seed=rep)
        t0 = time.perf_counter(); sc.render_device(p, tiles, n); dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    st = sc.stats()
    print(f"{name:14s} {W * H * spp / best / 1e6:8.1f} Msamples/s  {best * 1e3:7.1f} ms   segments/sample {st.segments / st.samples:.2f} shadow {st.shadow_rays / st.samples:.2f} sky {st.sky_fetches / st.samples:.2f}", flush=True)
    sc.close()
