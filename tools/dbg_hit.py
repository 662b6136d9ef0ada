import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lumillyrender_amd import device, host
from oracle import binding as oracle
from tests.test_gpu_parity import _random_rays, load
for name in ["cbox-spheres.toml", "brdf-row.toml", "two-spheres.toml"]:
    desc = load(name, 64, 64); scene = device.Scene(desc)
    o, d = _random_rays(desc, 20000, 3)
    gp, gt = scene.intersect(o, d); op, ot = oracle.intersect(desc, o, d)
    bad = np.nonzero((gp != op) | (gt != ot))[0]
    print(name, 'mismatches', len(bad))
    for i in bad[:12]:
        print('  ray', i, 'o', o[i], 'd', d[i], 'gpu', gp[i], gt[i], 'oracle', op[i], ot[i])
