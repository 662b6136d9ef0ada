"""Verdict r3, lever (a): would a SHARED-ORIGIN PACKET -- the direct-light connection and the continuation ray of one vertex walked as
one traversal of the tree, scene.rs:94-97,112-117 -- beat walking them one after the other in k_path_tree?

In SIMT a packet step executes both rays' slab tests (and both rays' primitive tests in a leaf) for every node of the UNION of the
two walks; separate walks execute one ray's tests for each node of each walk.  With the instruction counts of the 4-wide node step
of lr_path.h (ptrav_node: ~30 VALU per-origin + ~95 per-direction + ~25 ordering/push; ISA of round 4) a packet step costs
30 + 2 * 95 + 25 = 245 against 150, a leaf primitive 119 (flat_test_pair) against 2 * 60.  The packet wins on instructions iff
        U * 245 < (Nc + Nk) * 150        i.e.   U / (Nc + Nk) < 0.61
where Nc, Nk = nodes visited by the connection / the continuation ray alone and U = |nodes(conn) u nodes(cont)| (a lower bound of what
the packet visits: it orders children for one of the two rays only).  This script MEASURES that ratio on the stated scene:
vertices = first hits of camera rays (and of one diffuse bounce), connection = to a uniform point of the emitters, continuation = a
cosine-weighted direction; binary SAH tree of the description (what lr_scene_create collapses into 4-wide nodes), near-first walks
with distance culling (closest hit) resp. the bound dist + 2 EPS (connection, as ptrav_node), exact Moeller-Trumbore / sphere tests.

usage: python tools/pair_walk_overlap.py [scene.toml] [n_vertices]      (CPU only: host library + numpy)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lumillyrender_amd import host  # noqa: E402

EPS = 1e-3


def load(scene):
    d = host.Description(os.path.join(ROOT, "scenes", scene))
    desc = d.desc
    n = desc.n_prims
    prims = np.zeros((n, 9)); ptype = np.zeros(n, int); pmat = np.zeros(n, int)
    for i in range(n):
        p = desc.prims[i]
        prims[i] = list(p.v); ptype[i] = p.type; pmat[i] = p.material
    nn = desc.n_bvh_nodes
    boxes = np.zeros((nn, 2, 2, 3)); child = np.zeros((nn, 2), int)
    for i in range(nn):
        b = desc.bvh_nodes[i]
        for c in range(2):
            boxes[i, c, 0] = [b.x[2 * c], b.y[2 * c], b.z[2 * c]]
            boxes[i, c, 1] = [b.x[2 * c + 1], b.y[2 * c + 1], b.z[2 * c + 1]]
        child[i] = list(b.child)
    order = np.array([desc.bvh_prim_order[i] for i in range(n)])
    emit = [i for i in range(n) if any(desc.materials[pmat[i]].emission)]
    return d, prims, ptype, boxes, child, order, emit


def tri(p, o, d):
    p0, e1, e2 = p[0:3], p[3:6] - p[0:3], p[6:9] - p[0:3]
    pv = np.cross(d, e2); det = e1 @ pv
    if abs(det) < EPS: return None
    tv = o - p0; u = (tv @ pv) / det
    if u < 0 or u > 1: return None
    qv = np.cross(tv, e1); v = (d @ qv) / det
    if v < 0 or u + v > 1: return None
    t = (e2 @ qv) / det
    return t if t >= EPS else None


def sphere(p, o, d):
    co = o - p[0:3]; cod = co @ d; det = cod * cod - co @ co + p[3] * p[3]
    if det <= 0: return None
    s = np.sqrt(det); t1, t2 = -cod - s, -cod + s
    if t1 < EPS and t2 < EPS: return None
    return t1 if t1 > EPS else t2


def slab(box, o, inv, bound):
    t0 = (box[0] - o) * inv; t1 = (box[1] - o) * inv
    tn = max(np.minimum(t0, t1).max(), 0.0); tf = min(np.maximum(t0, t1).min(), bound)
    return tn if tn <= tf else None


def walk(sc, o, d, dist=None):
    """Near-first walk; returns (closest t, prim, set of inner nodes visited, primitives tested).  dist: a connection (bound dist + 2 EPS,
    stops at the first occluder)."""
    _, prims, ptype, boxes, child, order, _ = sc
    dd = np.where(np.abs(d) < 1e-20, np.copysign(1e-20, d), d); inv = 1.0 / dd
    best, bp = (np.inf, -1)
    visited, tested = set(), 0
    stack = [0]
    while stack:
        cur = stack.pop()
        if cur < 0:
            enc = ~cur; first, count = enc >> 3, enc & 7
            for k in range(first, first + count):
                i = order[k]; tested += 1
                t = tri(prims[i], o, d) if ptype[i] == 0 else sphere(prims[i], o, d)
                if t is None: continue
                if dist is not None:
                    if t - dist < -EPS: return t, i, visited, tested
                    if t - dist > EPS: continue
                if t < best or (t == best and i < bp): best, bp = t, i
            continue
        visited.add(cur)
        bound = dist + 2 * EPS if dist is not None else best
        hits = []
        for c in range(2):
            tn = slab(boxes[cur, c], o, inv, bound)
            if tn is not None: hits.append((tn, child[cur, c]))
        hits.sort(key=lambda h: -h[0])                     # far first on the stack
        for _, c in hits: stack.append(c)
    return best, bp, visited, tested


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "ibl-lens.toml"
    n_vert = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    sc = load(scene)
    d, prims, ptype, boxes, child, order, emit = sc
    cam = d.desc.camera
    rng = np.random.default_rng(7)
    pos = np.array(list(cam.aperture_position)); fwd = np.array(list(cam.forward)); right = np.array(list(cam.right)); up = np.array(list(cam.up))
    sw, sh = cam.sensor_size[0], cam.sensor_size[1]; asd = cam.aperture_sensor_distance
    rows = []
    tries = 0
    while len(rows) < n_vert and tries < 20 * n_vert:
        tries += 1
        px, py = (rng.random() - 0.5) * sw, (rng.random() - 0.5) * sh
        dirn = fwd * asd + right * px - up * py; dirn /= np.linalg.norm(dirn)
        o = pos
        for bounce in range(2):                                  # the camera vertex and one diffuse bounce
            t, p, _, _ = walk(sc, o, dirn)
            if p < 0: break
            x = o + dirn * t
            if ptype[p] == 0:
                nrm = np.cross(prims[p][3:6] - prims[p][0:3], prims[p][6:9] - prims[p][0:3]); nrm /= np.linalg.norm(nrm)
            else:
                nrm = (x - prims[p][0:3]) / prims[p][3]
            if nrm @ dirn > 0: nrm = -nrm
            # continuation: cosine-weighted about nrm
            a = np.array([0.0, 1.0, 0.0]) if abs(nrm[0]) > EPS else np.array([1.0, 0.0, 0.0])
            tx = np.cross(a, nrm); tx /= np.linalg.norm(tx); bx = np.cross(nrm, tx)
            r1, r2 = 2 * np.pi * rng.random(), rng.random()
            cont = tx * np.cos(r1) * np.sqrt(r2) + bx * np.sin(r1) * np.sqrt(r2) + nrm * np.sqrt(1 - r2)
            # connection: a uniform point on a random emitter triangle
            e = prims[emit[rng.integers(len(emit))]]
            u, v = rng.random(), rng.random(); mn, mx = min(u, v), max(u, v)
            lp = e[0:3] * mn + e[3:6] * (1 - mx) + e[6:9] * (mx - mn)
            dp = lp - x; dist = np.linalg.norm(dp); cdir = dp / dist
            if cdir @ nrm > 0:
                _, _, vc, tc = walk(sc, x, cdir, dist)
                _, _, vk, tk = walk(sc, x, cont)
                rows.append((len(vc), len(vk), len(vc | vk), tc, tk, bounce))
            o, dirn = x, cont
    r = np.array(rows, float)
    nc, nk, u = r[:, 0].sum(), r[:, 1].sum(), r[:, 2].sum()
    print(f"scene {scene}: {len(r)} vertices with a connection ({int((r[:, 5] == 0).sum())} camera vertices, {int((r[:, 5] == 1).sum())} after one bounce)")
    print(f"binary inner nodes per walk: connection {r[:, 0].mean():.1f}, continuation {r[:, 1].mean():.1f}, union {r[:, 2].mean():.1f}")
    print(f"U / (Nc + Nk) = {u / (nc + nk):.3f}   (shared nodes: {(nc + nk - u) / (nc + nk):.3f} of the separate walks' visits; a packet wins on VALU instructions below 0.61)")
    print(f"primitives tested per walk: connection {r[:, 3].mean():.2f}, continuation {r[:, 4].mean():.2f}")
    print(f"packet / separate instruction estimate: node steps {u * 245 / ((nc + nk) * 150):.2f}x")


if __name__ == "__main__":
    main()
