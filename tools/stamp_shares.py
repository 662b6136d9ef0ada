"""Phase shares of the resident pipeline (k_resident) from the stamp build (make -C lumillyrender_amd/csrc stamp -> build/v_stamp.so): wave cycles in
trace / shade / shadow+finish and parked at each of the three workgroup barriers, and the part of the LAST barrier's wait that came after every one
of the wave's own 64 slots had completed phase 3 -- the most that per-slot ready flags in place of that barrier could return.
usage (GPU box): python tools/stamp_shares.py [scene W H spp]...     default: the three BRDF rows at 960x540, 512 spp"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
from lumillyrender_amd import device, host
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = host.Description(os.path.join(%r, "scenes", name)); d.set_resolution(W, H)
sc = device.Scene(d)
tiles, n = host.full_tile(W, H)
sc.render_device(d.render_params(spp=spp, seed=1), tiles, n)
st = sc.stats()
print("render_ms %%.2f  Msamples/s %%.0f" %% (st.render_ms, W * H * spp / st.render_ms / 1e3))
''' % (ROOT, ROOT)

def main():
    args = sys.argv[1:]
    wls = [args[i:i + 4] for i in range(0, len(args), 4)] or [["brdf-row.toml", "960", "540", "512"], ["brdf-row-phong.toml", "960", "540", "512"],
                                                              ["brdf-row-blinn-phong.toml", "960", "540", "512"]]
    env = dict(os.environ, LR_HIP_LIB=os.path.join(ROOT, "build", "v_stamp.so"))
    for wl in wls:
        out = subprocess.run([sys.executable, "-c", CHILD] + wl, env=env, capture_output=True, text=True, timeout=900)
        print("==", " ".join(wl), "rc", out.returncode)
        print(out.stdout.strip())
        for l in out.stderr.splitlines():
            if "LR_STAMP" in l or out.returncode: print(l)

if __name__ == "__main__":
    main()
