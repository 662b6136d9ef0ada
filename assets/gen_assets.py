"""Procedural stand-ins for the assets the reference scenes point at but never published
(SURVEY.md 5.9): a closed 100 000-triangle mesh for models/bunny/bunny.obj and a 3072x1536
equirectangular HDR environment for models/ibl/*.hdr.  Deterministic (fixed seeds); outputs are
git-ignored and recreated by __graft_entry__.build().

    python assets/gen_assets.py [--force]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BLOB = os.path.join(HERE, "models", "blob", "blob.obj")
SKY = os.path.join(HERE, "models", "ibl", "sky_3k.hdr")


def make_blob(path, n_lon=250, n_lat=201, seed=1):
    """Displaced UV sphere: 2 * n_lon * (n_lat - 1) = 100 000 triangles, radius ~0.75, sitting on y = 0.03
    (the size and placement of the bunny the scenes scale by 130)."""
    rng = np.random.default_rng(seed)
    waves = [(rng.integers(1, 6), rng.integers(1, 6), rng.random() * 2 * np.pi, rng.random() * 2 * np.pi, 0.04 + 0.05 * rng.random()) for _ in range(7)]
    verts = []
    def radius(theta, phi):
        r = 0.75
        for a, b, p0, p1, amp in waves:
            r += amp * np.sin(a * theta + p0) * np.cos(b * phi + p1) * np.sin(theta)
        return r
    verts.append((0.0, 0.78 + radius(0.0, 0.0), 0.0))                      # north pole
    for i in range(1, n_lat):
        theta = np.pi * i / n_lat
        for j in range(n_lon):
            phi = 2 * np.pi * j / n_lon
            r = radius(theta, phi)
            verts.append((r * np.sin(theta) * np.cos(phi), 0.78 + r * np.cos(theta), r * np.sin(theta) * np.sin(phi)))
    verts.append((0.0, 0.78 - radius(np.pi, 0.0), 0.0))                    # south pole
    south = len(verts)
    faces = []
    ring = lambda i, j: 2 + (i - 1) * n_lon + (j % n_lon)                  # 1-based index of ring i (1..n_lat-1)
    for j in range(n_lon):
        faces.append((1, ring(1, j + 1), ring(1, j)))
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j + 1), ring(i + 1, j)
            faces.append((a, b, c)); faces.append((a, c, d))
    for j in range(n_lon):
        faces.append((south, ring(n_lat - 1, j), ring(n_lat - 1, j + 1)))
    assert len(faces) == 2 * n_lon * (n_lat - 1)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("# procedural displaced sphere, %d triangles (assets/gen_assets.py, seed %d)\n" % (len(faces), seed))
        f.write("g blob\n")
        for v in verts:
            f.write("v %.6f %.6f %.6f\n" % v)
        for a, b, c in faces:
            f.write("f %d %d %d\n" % (a, b, c))
    return len(faces)


def make_sky(path, height=1536, seed=2):
    sys.path.insert(0, ROOT)
    from lumillyrender_amd import host
    rng = np.random.default_rng(seed)
    w = 2 * height
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(w) + 0.5) / w
    theta = v[:, None] * np.pi
    phi = u[None, :] * 2 * np.pi
    up = np.cos(theta)
    horizon = np.exp(-(up * 3.0) ** 2)
    img = np.empty((height, w, 3), dtype=np.float32)
    sky = np.array([0.35, 0.55, 0.95]); ground = np.array([0.25, 0.22, 0.2]); haze = np.array([0.9, 0.85, 0.8])
    t = np.clip(up * 0.5 + 0.5, 0, 1)
    for c in range(3):
        img[..., c] = (ground[c] * (1 - t) + sky[c] * t) * (1 - 0.6 * horizon) + haze[c] * 0.6 * horizon + 0 * phi
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) + 0 * phi, np.sin(theta) * np.sin(phi)], axis=-1)
    for _ in range(5):                                                      # a few ~10^3-intensity blobs
        dirv = rng.standard_normal(3); dirv[1] = abs(dirv[1]) + 0.3; dirv /= np.linalg.norm(dirv)
        cosang = np.clip(d @ dirv, -1, 1)
        sharp = 400 + 3000 * rng.random()
        col = np.array([1.0, 0.9 + 0.1 * rng.random(), 0.7 + 0.3 * rng.random()]) * (300 + 1500 * rng.random())
        img += (np.exp((cosang - 1) * sharp))[..., None].astype(np.float32) * col.astype(np.float32)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    host.save_hdr(path, img)


if __name__ == "__main__":
    force = "--force" in sys.argv
    if force or not os.path.exists(BLOB):
        n = make_blob(BLOB)
        print("wrote", BLOB, n, "triangles")
    if force or not os.path.exists(SKY):
        make_sky(SKY)
        print("wrote", SKY)
