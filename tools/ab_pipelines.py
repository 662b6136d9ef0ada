"""A/B of the pipelines in ONE process, interleaved rounds (cdna guide rule 24): Msamples/s of lr_render_device per variant.
usage: ab_pipelines.py scene W H spp rounds variant[,variant...]      variant = auto | fused | resident | streaming"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lumillyrender_amd import abi, device, host
name, W, H, spp, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
variants = sys.argv[6].split(",")
FLAGS = {"auto": 0, "fused": abi.LR_FLAG_FUSED, "resident": abi.LR_FLAG_RESIDENT, "streaming": abi.LR_FLAG_STREAMING}
d = host.Description(os.path.join(ROOT, "scenes", name)); d.set_resolution(W, H)
sc = device.Scene(d)
tiles, n = host.full_tile(W, H)
res = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        p = d.render_params(spp=spp, seed=r, flags=FLAGS[v] | abi.LR_FLAG_PROFILE)
        t0 = time.perf_counter(); sc.render_device(p, tiles, n); dt = time.perf_counter() - t0
        st = sc.stats()
        if r: res[v].append((W * H * spp / dt / 1e6, st.render_ms, int(st.pipeline), int(st.samples)))
for v in variants:
    rates = sorted(x[0] for x in res[v])
    print(json.dumps({"variant": v, "scene": name, "W": W, "H": H, "spp": spp, "pipeline": res[v][0][2], "samples_ok": all(x[3] == W * H * spp for x in res[v]),
                      "Msamples_s_median": round(rates[len(rates) // 2], 1), "min": round(rates[0], 1), "max": round(rates[-1], 1),
                      "device_ms_min": round(min(x[1] for x in res[v]), 3)}), flush=True)
