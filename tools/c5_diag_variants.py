"""LR_DIAG phase shares of config 5 with one feature replaced at a time (run with LR_HIP_LIB=build/v_diag.so): where the GGX mesh's and the
connections' time sits inside k_path_tree.  usage: LR_HIP_LIB=$PWD/build/v_diag.so python tools/c5_diag_variants.py [W H spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lumillyrender_amd import device, host
W, H, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 2048, 64)
text = open(os.path.join(ROOT, "scenes", "ibl-lens.toml")).read()
variants = {
    "as stated": lambda t: t,
    "lambert mesh": lambda t: t.replace('mesh = "blob"\nmaterial = "gloss"', 'mesh = "blob"\nmaterial = "matte"'),
    "uniform sky": lambda t: t.replace('type = "ibl"\npath = "models/ibl/sky_3k.hdr"', 'type = "uniform"\ncolor = [1, 1, 1]'),
    "pt": lambda t: t.replace('integrator = "pt-direct"', 'integrator = "pt"'),
}
for name, edit in variants.items():
    d = host.Description(text=edit(text)); d.set_resolution(W, H)
    sc = device.Scene(d)
    tiles, n = host.full_tile(W, H)
    print(f"=== {name}", file=sys.stderr, flush=True)
    sc.render_device(d.render_params(spp=spp, seed=1), tiles, n)
    st = sc.stats()
    print(f"    {W * H * spp / st.render_ms / 1e3:.1f} Msamples/s (diag build)  segments/sample {st.segments / st.samples:.2f}", file=sys.stderr, flush=True)
    sc.close()
