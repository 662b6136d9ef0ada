"""Interleaved A/B of library builds / environment switches on one GPU box: every round runs every variant once, each in a fresh
process (one library per process), so clock and thermal drift hit all variants alike.
usage: ab4.py "<scene W H spp>[;<scene W H spp>...]" rounds variant [variant ...]
  variant = path of a liblumilly_hip.so build ("product" = the in-tree one), optionally followed by ,NAME=VALUE environment switches
Prints the median / min / max device milliseconds (lr_get_stats.render_ms: the render's own HIP events) and Msamples/s per variant."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
from lumillyrender_amd import abi, device, host
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = host.Description(os.path.join(%r, "scenes", name)); d.set_resolution(W, H)
sc = device.Scene(d)
tiles, n = host.full_tile(W, H)
best = 1e30
for rep in range(3):
    p = d.render_params(spp=spp if rep else max(1, spp // 8), seed=rep)
    sc.render_device(p, tiles, n)
    st = sc.stats()
    if rep:
        assert st.samples == W * H * spp
        best = min(best, st.render_ms)
print(json.dumps({"ms": best}))
''' % (ROOT, ROOT)

def main():
    workloads = [w.split() for w in sys.argv[1].split(";")]
    rounds = int(sys.argv[2])
    variants = sys.argv[3:]
    res = {(i, v): [] for i in range(len(workloads)) for v in variants}
    for r in range(rounds):
        for i, wl in enumerate(workloads):
            for v in variants:
                parts = v.split(",")
                env = dict(os.environ)
                if parts[0] != "product":
                    env["LR_HIP_LIB"] = os.path.join(ROOT, parts[0])
                for kv in parts[1:]:
                    k, val = kv.split("=", 1); env[k] = val
                out = subprocess.run([sys.executable, "-c", CHILD] + wl, env=env, capture_output=True, text=True, timeout=900)
                if out.returncode != 0:
                    print(f"!! {v} {wl}: rc {out.returncode}: {out.stderr[-400:]}", flush=True); continue
                res[(i, v)].append(json.loads(out.stdout.strip().splitlines()[-1])["ms"])
    for i, wl in enumerate(workloads):
        W, H, spp = int(wl[1]), int(wl[2]), int(wl[3])
        for v in variants:
            ms = sorted(res[(i, v)])
            if not ms: continue
            med = ms[len(ms) // 2]
            print(json.dumps({"workload": " ".join(wl), "variant": v, "rounds": len(ms), "device_ms_median": round(med, 3), "min": round(ms[0], 3), "max": round(ms[-1], 3),
                              "Msamples_s_median": round(W * H * spp / med / 1e3, 1)}), flush=True)

if __name__ == "__main__":
    main()
