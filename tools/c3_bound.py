"""Verdict r3 item 4: would a class exchange (vertices of one BSDF class gathered into one wave through LDS, path state staying in the
lanes) lift config 3 to 11.5 Gsamples/s?  A bound that needs no new kernel: the fused kernel on the BRDF row with EVERY material
made Lambert (no wave ever runs two BSDF bodies: what a free, perfect exchange would approach for the Lambert share of the vertices)
and with every material made GGX (the GGX share).  Path lengths change with the materials, so rates are also reported per path
vertex.  usage: c3_bound.py [W H spp]"""
import os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lumillyrender_amd import abi, device, host
W, H, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (960, 540, 1024)
text = open(os.path.join(ROOT, "scenes", "brdf-row.toml")).read()


def all_lambert(t):
    t = re.sub(r'type = "ggx"\nreflectance = (\[[^\]]*\])\nroughness = [0-9.]+\nior = [0-9.]+', r'type = "lambert"\nalbedo = [0.9, 0.9, 0.9]', t)
    assert '"ggx"' not in t
    return t


def all_ggx(t):
    t = re.sub(r'type = "lambert"\nalbedo = (\[[^\]]*\])', r'type = "ggx"\nreflectance = \1\nroughness = 0.6\nior = 100000', t)
    # objects without a material (the floor under the spheres, the strip light) keep the loader's default Lambert: the light must stay Lambert (only Lambert emits)
    return t


variants = {"as stated": (lambda t: t), "all Lambert": all_lambert, "all GGX (but the emitter and the default floor)": all_ggx}
for name, edit in variants.items():
    d = host.Description(text=edit(text)); d.set_resolution(W, H)
    sc = device.Scene(d)
    tiles, n = host.full_tile(W, H)
    for label, flags in (("default", 0), ("fused", abi.LR_FLAG_FUSED), ("resident", abi.LR_FLAG_RESIDENT)):
        best, st = 1e9, None
        for rep in range(3):
            p = d.render_params(spp=spp, seed=rep, flags=flags)
            sc.render_device(p, tiles, n)
            st = sc.stats()
            if rep: best = min(best, st.render_ms)
        verts = st.segments
        print(f"{name:48s} {label:8s} pipeline {int(st.pipeline)}  {W * H * spp / best / 1e3:8.1f} Msamples/s  {best:7.1f} ms  "
              f"segments/sample {st.segments / st.samples:.2f}  shadow {st.shadow_rays / st.samples:.2f}  {verts / best / 1e6:6.2f} G vertices/s", flush=True)
    sc.close()
