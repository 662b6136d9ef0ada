"""Verdict r4, lever (b): would 8-WIDE compressed nodes with octant-ordered traversal (Ylitie, Karras, Laine 2017: one node =
8 quantised child boxes, child slots assigned at build time so that `slot XOR ray octant` is a front-to-back order -- no sorting
network, one stack entry per node = {node, hit mask}) need fewer VALU instructions in the node phase of k_path_tree than the
4-wide node step it has (ptrav_node: ~150 VALU per step with its 5-comparator sort and three LDS pushes)?

Priced the way round 4 priced the packet walk: COUNT the steps of both traversals on the stated scenes with host code, multiply by an
instruction model.
  4-wide (as built)   near-first: the hit children sorted by entry distance, nearest next, the others pushed far-first; a popped child
                      is entered without a re-test (its box was tested against the bound of the time); distance culling by the
                      closest hit so far (closest-hit query) or dist + 2 EPS (connection).  ~150 VALU per node step.
  8-wide octant       the hit children of a node go on the stack as ONE entry; they are entered in `slot XOR octant` order; entry
                      distances are not kept, so a child whose box lies behind a hit found meanwhile is still entered and costs a
                      node step (its own children then fail the shortened slab test).  Model: 30 (ray -> node grid) + 8 x 18 (slab
                      tests) + 16 (hit mask, octant permutation, push) = 190 VALU per node step + 12 per child taken from a mask.
Both trees are collapsed from the SAME binary SAH tree of the description by the rule lr_scene_create uses (a node adopts the
children of its largest inner child until its slots are full) and both store child boxes on an 8-bit grid over the node's union,
rounded outward (the 8-wide node's grid is coarser: its union is larger).  Primitive tests are counted too (an order that is only
approximately front-to-back finds the closest hit later and tests more primitives: 53 VALU per triangle).

usage: python tools/wide_node_model.py [scene.toml] [n_vertices] [out.json]      (CPU only: host library + numpy + scipy)"""
import json
import os
import sys

import numpy as np
from scipy.optimize import linear_sum_assignment

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import pair_walk_overlap as pw  # noqa: E402  (scene loading, exact primitive tests, the binary near-first walk)

EPS = pw.EPS
VALU_4 = 150.0
VALU_8_STEP, VALU_8_POP = 190.0, 12.0
VALU_TRI, VALU_LEAF = 53.0, 20.0


class Wide:
    """A W-wide tree collapsed from the binary one: per node `refs` (child node index >= 0, leaf ~enc < 0) and quantised boxes."""

    def __init__(self, boxes, child, width, octant_slots):
        self.width, self.nodes = width, []
        self.octant_slots = octant_slots
        self._boxes, self._child = boxes, child
        self._build(0)

    def _kids(self, node):
        return [(self._boxes[node, c, 0], self._boxes[node, c, 1], int(self._child[node, c])) for c in range(2)]

    @staticmethod
    def _area(lo, hi):
        d = hi - lo
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]

    def _build(self, node):
        me = len(self.nodes)
        self.nodes.append(None)
        c = self._kids(node)
        while len(c) < self.width:
            pick, best = -1, -1.0
            for k, (lo, hi, ref) in enumerate(c):
                if ref >= 0 and self._area(lo, hi) > best:
                    best, pick = self._area(lo, hi), k
            if pick < 0:
                break
            two = self._kids(c[pick][2])
            c[pick] = two[0]; c.append(two[1])
        lo = np.min([k[0] for k in c], axis=0); hi = np.max([k[1] for k in c], axis=0)
        ext = np.maximum(hi - lo, 1e-30)
        step = 2.0 ** np.ceil(np.log2(ext / 255.0))
        qlo = [lo + np.floor((k[0] - lo) / step) * step for k in c]
        qhi = [lo + np.ceil((k[1] - lo) / step) * step for k in c]
        slots = list(range(len(c)))
        if self.octant_slots and len(c) > 1:
            # Ylitie et al. section 3.2: assign children to slots so that, for the ray octant s, visiting slots in the order of
            # `slot XOR s` descending is front-to-back: cost(child, slot) = (centroid - node centre) . (sign vector of the slot)
            ctr = 0.5 * (lo + hi)
            cost = np.zeros((len(c), self.width))
            for i, k in enumerate(c):
                v = 0.5 * (k[0] + k[1]) - ctr
                for s in range(self.width):
                    sign = np.array([1.0 if (s >> a) & 1 else -1.0 for a in range(3)])
                    cost[i, s] = v @ sign
            rows, cols = linear_sum_assignment(-cost)
            slots = [int(cols[list(rows).index(i)]) for i in range(len(c))]
        refs = []
        for i, k in enumerate(c):
            ref = k[2]
            refs.append([qlo[i], qhi[i], ref, slots[i]])
        self.nodes[me] = refs
        for r in refs:
            if r[2] >= 0:
                r[2] = self._build(r[2])
        return me


def slab(lo, hi, o, inv, bound):
    t0 = (lo - o) * inv; t1 = (hi - o) * inv
    tn = max(np.minimum(t0, t1).max(), 0.0); tf = min(np.maximum(t0, t1).min(), bound)
    return tn if tn <= tf else None


def leaf(sc, ref, o, d, dist, best, bp):
    _, prims, ptype, _, _, order, _ = sc
    enc = ~ref; first, count = enc >> 3, enc & 7
    tested, occluded = 0, False
    for k in range(first, first + count):
        i = order[k]; tested += 1
        t = pw.tri(prims[i], o, d) if ptype[i] == 0 else pw.sphere(prims[i], o, d)
        if t is None:
            continue
        if dist is not None:
            if t - dist < -EPS:
                occluded = True; break
            if t - dist > EPS:
                continue
        if t < best or (t == best and i < bp):
            best, bp = t, i
    return best, bp, tested, occluded


def walk_sorted(sc, tree, o, d, dist=None):
    """the 4-wide walk of ptrav_node / ptrav_leaf"""
    dd = np.where(np.abs(d) < 1e-20, np.copysign(1e-20, d), d); inv = 1.0 / dd
    best, bp, steps, tests, leaves = np.inf, -1, 0, 0, 0
    stack = [0]
    while stack:
        cur = stack.pop()
        if cur < 0:
            best, bp, t, occ = leaf(sc, cur, o, d, dist, best, bp); tests += t; leaves += 1
            if occ:
                break
            continue
        steps += 1
        bound = dist + 2 * EPS if dist is not None else best
        hits = []
        for lo, hi, ref, _ in tree.nodes[cur]:
            tn = slab(lo, hi, o, inv, bound)
            if tn is not None:
                hits.append((tn, ref))
        hits.sort(key=lambda h: -h[0])
        for _, r in hits:
            stack.append(r)
    return steps, 0, tests, leaves


def walk_octant(sc, tree, o, d, dist=None):
    """the 8-wide walk: one stack entry per node, children entered in slot-XOR-octant order, no entry distances kept"""
    dd = np.where(np.abs(d) < 1e-20, np.copysign(1e-20, d), d); inv = 1.0 / dd
    # a slot's sign vector points from the node's centre to its child; the slot whose signs are the OPPOSITE of the ray's is the
    # nearest, so with bit a of `octant` set where d[a] < 0, slot XOR octant = 0 is nearest and 7 farthest
    octant = (1 if d[0] < 0 else 0) | (2 if d[1] < 0 else 0) | (4 if d[2] < 0 else 0)
    best, bp, steps, pops, tests, leaves = np.inf, -1, 0, 0, 0, 0
    stack = [[(0, 0)]]                                                 # groups of (priority, ref), nearest = largest priority last
    while stack:
        grp = stack[-1]
        _, cur = grp.pop()
        if not grp:
            stack.pop()
        pops += 1
        if cur < 0:
            best, bp, t, occ = leaf(sc, cur, o, d, dist, best, bp); tests += t; leaves += 1
            if occ:
                break
            continue
        steps += 1
        bound = dist + 2 * EPS if dist is not None else best
        hits = []
        for lo, hi, ref, slot in tree.nodes[cur]:
            if slab(lo, hi, o, inv, bound) is not None:
                hits.append(((slot ^ octant) & 7, ref))
        if hits:
            hits.sort(key=lambda h: -h[0])                             # entered from the end: smallest slot XOR octant (nearest) first
            stack.append(hits)
    return steps, pops, tests, leaves


def vertices(sc, d, n_vert, rng, nee):
    """path vertices of the scene's camera (first hit + two diffuse bounces): (origin, continuation dir, connection dir, dist)"""
    _, prims, ptype, boxes, child, order, emit = sc
    cam = d.desc.camera
    pos = np.array(list(cam.aperture_position)); fwd = np.array(list(cam.forward)); right = np.array(list(cam.right)); up = np.array(list(cam.up))
    sw, sh = cam.sensor_size[0], cam.sensor_size[1]; asd = cam.aperture_sensor_distance
    out, tries = [], 0
    while len(out) < n_vert and tries < 20 * n_vert:
        tries += 1
        px, py = (rng.random() - 0.5) * sw, (rng.random() - 0.5) * sh
        dirn = fwd * asd + right * px - up * py; dirn /= np.linalg.norm(dirn)
        o = pos
        out.append((o, dirn, None, None))                              # the camera ray itself is a closest-hit query
        for bounce in range(3):
            t, p, _, _ = pw.walk(sc, o, dirn)
            if p < 0:
                break
            x = o + dirn * t
            if ptype[p] == 0:
                nrm = np.cross(prims[p][3:6] - prims[p][0:3], prims[p][6:9] - prims[p][0:3]); nrm /= np.linalg.norm(nrm)
            else:
                nrm = (x - prims[p][0:3]) / prims[p][3]
            if nrm @ dirn > 0:
                nrm = -nrm
            a = np.array([0.0, 1.0, 0.0]) if abs(nrm[0]) > EPS else np.array([1.0, 0.0, 0.0])
            tx = np.cross(a, nrm); tx /= np.linalg.norm(tx); bx = np.cross(nrm, tx)
            r1, r2 = 2 * np.pi * rng.random(), rng.random()
            cont = tx * np.cos(r1) * np.sqrt(r2) + bx * np.sin(r1) * np.sqrt(r2) + nrm * np.sqrt(1 - r2)
            cdir, dist = None, None
            if nee and emit:
                e = prims[emit[rng.integers(len(emit))]]
                u, v = rng.random(), rng.random(); mn, mx = min(u, v), max(u, v)
                lp = e[0:3] * mn + e[3:6] * (1 - mx) + e[6:9] * (mx - mn)
                dp = lp - x; dist = float(np.linalg.norm(dp)); cdir = dp / dist
                if cdir @ nrm <= 0:
                    cdir, dist = None, None
            out.append((x, cont, cdir, dist))
            o, dirn = x, cont
    return out[:n_vert]


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "mesh-box.toml"
    n_vert = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
    sc = pw.load(scene)
    d, prims, ptype, boxes, child, order, emit = sc
    nee = d.renderer.integrator == 1
    t4 = Wide(boxes, child, 4, False)
    t8 = Wide(boxes, child, 8, True)
    t8s = Wide(boxes, child, 8, False)
    rng = np.random.default_rng(7)
    vs = vertices(sc, d, n_vert, rng, nee)
    acc = {k: np.zeros(4) for k in ("w4", "w8_octant", "w8_sorted")}
    n_walks = 0
    for (o, cont, cdir, dist) in vs:
        for (dirn, dd) in ((cont, None), (cdir, dist)):
            if dirn is None:
                continue
            n_walks += 1
            acc["w4"] += walk_sorted(sc, t4, o, dirn, dd)
            acc["w8_octant"] += walk_octant(sc, t8, o, dirn, dd)
            acc["w8_sorted"] += walk_sorted(sc, t8s, o, dirn, dd)
    res = {"scene": scene, "walks": n_walks, "vertices": len(vs), "nodes_4wide": len(t4.nodes), "nodes_8wide": len(t8.nodes)}
    for k, v in acc.items():
        steps, pops, tests, leaves = v / n_walks
        valu_node = steps * VALU_4 if k == "w4" else steps * VALU_8_STEP + pops * VALU_8_POP
        if k == "w8_sorted":
            valu_node = steps * (30 + 8 * 18 + 19 * 5 + 30)            # a 19-comparator network on (key, ref) pairs + up to seven pushes
        res[k] = {"node_steps_per_walk": round(steps, 2), "children_popped_per_walk": round(pops, 2), "primitive_tests_per_walk": round(tests, 2),
                  "leaves_per_walk": round(leaves, 2), "valu_node_phase": round(valu_node, 1), "valu_leaf_phase": round(tests * VALU_TRI + leaves * VALU_LEAF, 1)}
    base = res["w4"]["valu_node_phase"] + res["w4"]["valu_leaf_phase"]
    for k in ("w8_octant", "w8_sorted"):
        res[k]["node_phase_vs_4wide"] = round(res[k]["valu_node_phase"] / res["w4"]["valu_node_phase"], 3)
        res[k]["walk_vs_4wide"] = round((res[k]["valu_node_phase"] + res[k]["valu_leaf_phase"]) / base, 3)
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
