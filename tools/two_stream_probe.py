"""Does the streaming pipeline gain from two half-frame renders running concurrently on two HIP streams (trace of one
overlapping the bandwidth-bound shade of the other)?  usage: two_stream_probe.py scene W H spp"""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from lumillyrender_amd import host, device
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = host.Description("scenes/" + name); d.set_resolution(W, H)
one = device.Scene(d)
tiles, n = host.full_tile(W, H)
for rep in range(2):
    t0 = time.time(); one.render_device(d.render_params(spp=spp, seed=rep), tiles, n); t1 = time.time()
print("one handle, full frame: %.1f Msamples/s" % (W * H * spp / (t1 - t0) / 1e6))
one.close()
scenes = [device.Scene(d) for _ in range(2)]
shards = [host.tiles(W, H, 64, r, 2) for r in range(2)]
def work(r, seed):
    scenes[r].render_device(d.render_params(spp=spp, seed=seed), *shards[r])
for rep in range(2):
    ts = [threading.Thread(target=work, args=(r, rep)) for r in range(2)]
    t0 = time.time(); [t.start() for t in ts]; [t.join() for t in ts]; t1 = time.time()
print("two handles, half the tiles each, concurrent: %.1f Msamples/s" % (W * H * spp / (t1 - t0) / 1e6))
