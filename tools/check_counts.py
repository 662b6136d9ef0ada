"""usage: check_counts.py scene W H spp slots flags -- renders and compares the device's finished-sample counter and the
film's mean with the nominal W*H*spp (a lost or duplicated work item shows up in both)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from lumillyrender_amd import host, device
name, W, H, spp, slots, flags = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
d = host.Description("scenes/" + name); d.set_resolution(W, H)
sc = device.Scene(d)
img = sc.render(d.render_params(spp=spp, seed=1, flags=flags, path_slots=slots))
st = sc.stats()
print(name, W, H, spp, "slots", slots, "samples", st.samples, "nominal", W * H * spp, "ratio", st.samples / (W * H * spp), "iterations", st.iterations,
      "film mean", float(img.mean()), "zero pixels", int((img.sum(axis=2) == 0).sum()), "nan", int(np.isnan(img).sum()))
