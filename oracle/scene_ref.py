"""Python (tomli + numpy float32) restatement of the reference's scene front end -- TEST
INFRASTRUCTURE ONLY (checks the C++ host loader in lumillyrender_amd/host/).

Follows, with f32 arithmetic in the reference's operation order:
  scene_loader.rs:8-270   schema, defaults, name lookup, light -> object binding, transform fold
  math/matrix4.rs:9-68,193-222   unit/translate/scale/axis_angle/look_at, M*v, M*M (row-major)
  camera.rs:34-62, :148-166, :366-409   camera constructors
  description.rs:89-197   instantiation order, sphere centre/radius, OBJ faces (fan triangulation)
"""
import math
import os

import numpy as np
import tomli

F = np.float32
PI = F(3.14159265358979323846264338327950288)


def v3(x):
    return np.array([F(float(c)) for c in x], dtype=np.float32)


def dot4(r, x, y, z, w):
    return F(F(F(F(r[0] * x) + F(r[1] * y)) + F(r[2] * z)) + F(r[3] * w))


def mat_vec(m, p):
    return np.array([dot4(m[4 * i:4 * i + 4], p[0], p[1], p[2], F(1.0)) for i in range(3)], dtype=np.float32)


def mat_mul(a, b):
    r = np.zeros(16, dtype=np.float32)
    for y in range(4):
        for x in range(4):
            r[4 * y + x] = F(F(F(F(a[4 * y] * b[x]) + F(a[4 * y + 1] * b[4 + x])) + F(a[4 * y + 2] * b[8 + x])) + F(a[4 * y + 3] * b[12 + x]))
    return r


def dot3(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def cross(a, b):
    return np.array([F(F(a[1] * b[2]) - F(a[2] * b[1])), F(F(a[2] * b[0]) - F(a[0] * b[2])), F(F(a[0] * b[1]) - F(a[1] * b[0]))], dtype=np.float32)


def normalize(a):
    n = F(np.sqrt(dot3(a, a)))
    return np.array([F(a[0] / n), F(a[1] / n), F(a[2] / n)], dtype=np.float32)


def transform_matrix(t):
    ty = t["type"]
    one, zero = F(1), F(0)
    if ty == "translate":
        v = v3(t["vector"])
        return np.array([1, 0, 0, v[0], 0, 1, 0, v[1], 0, 0, 1, v[2], 0, 0, 0, 1], dtype=np.float32)
    if ty == "scale":
        v = v3(t["vector"])
        return np.array([v[0], 0, 0, 0, 0, v[1], 0, 0, 0, 0, v[2], 0, 0, 0, 0, 1], dtype=np.float32)
    if ty == "axis-angle":
        a = v3(t["axis"])
        ang = F(F(F(float(t["angle"])) * PI) / F(180.0))
        c, s = F(math.cos(float(ang))), F(math.sin(float(ang)))
        # libm cosf/sinf of the f32 angle: evaluate in double and round (agrees with glibc to <1 ulp;
        # the comparison below uses a tolerance for the rotated entries)
        omc = F(one - c)
        return np.array([
            F(c + F(F(a[0] * a[0]) * omc)), F(F(F(a[0] * a[1]) * omc) - F(a[2] * s)), F(F(F(a[0] * a[2]) * omc) + F(a[1] * s)), zero,
            F(F(F(a[1] * a[0]) * omc) + F(a[2] * s)), F(c + F(F(a[1] * a[1]) * omc)), F(F(F(a[1] * a[2]) * omc) - F(a[0] * s)), zero,
            F(F(F(a[2] * a[0]) * omc) - F(a[1] * s)), F(F(F(a[2] * a[1]) * omc) + F(a[0] * s)), F(c + F(F(a[2] * a[2]) * omc)), zero,
            zero, zero, zero, one], dtype=np.float32)
    if ty == "look-at":
        origin, target, up = v3(t["origin"]), v3(t["target"]), v3(t["up"])
        za = normalize((origin - target).astype(np.float32))
        xa = normalize(cross(up, za))
        ya = cross(za, xa)
        return np.array([xa[0], xa[1], xa[2], 0, ya[0], ya[1], ya[2], 0, za[0], za[1], za[2], 0, origin[0], origin[1], origin[2], 1], dtype=np.float32)
    raise ValueError(ty)


def compose(ts):
    p = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1], dtype=np.float32)
    for t in ts:
        p = mat_mul(transform_matrix(t), p)
    return p


def camera(cfg, width, height):
    cam = cfg["camera"]
    m = compose(cam.get("transform", []))
    out = {"type": {"ideal-pinhole": 0, "thin-lens": 1, "omnidirectional": 2}[cam["type"]]}
    ap = m[12:15].copy()
    fwd, right, up = mat_vec(m, v3([0, 0, -1])), mat_vec(m, v3([1, 0, 0])), mat_vec(m, v3([0, 1, 0]))
    out.update(forward=fwd, right=right, up=up, aperture_position=ap, sensor_sensitivity=F(1.0))
    if out["type"] == 2:
        out["position"] = ap
        return out
    direction = (fwd * F(50.0)).astype(np.float32)
    out["position"] = (ap - direction).astype(np.float32)
    dist = F(np.sqrt(dot3(direction, direction)))
    out["aperture_sensor_distance"] = dist
    half = F(F(F(F(float(cam["fov"])) * PI) / F(180.0)) / F(2.0))
    sx = F(F(F(2.0) * dist) * F(math.tan(float(half))))
    sy = F(F(sx * F(height)) / F(width))
    out["sensor_size"] = np.array([sx, sy], dtype=np.float32)
    if out["type"] == 1:
        fd = F(float(cam.get("focus-distance", cam.get("focus_distance"))))
        fn = F(float(cam.get("f-number", cam.get("f_number"))))
        focal = F(F(1.0) / F(F(F(1.0) / dist) + F(F(1.0) / fd)))
        radius = F(F(focal / fn) / F(2.0))
        area = F(F(sx * sy) / F(width * height))
        out.update(aperture_radius=radius, focus_distance=fd, sensor_pixel_area=area,
                   sensor_sensitivity=F(F(dist * dist) / F(F(F(area * PI) * radius) * radius)))
    return out


def load_obj(path):
    """tobj-like: models split at o/g and at usemtl changes; fan triangulation; Kd from mtllib."""
    pos, models, mats, mat_map = [], [], [], {}
    cur, tmp, mat_id = "unnamed_object", [], -1

    def flush():
        models.append({"name": cur, "tris": list(tmp), "material_id": mat_id})
        tmp.clear()
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        k, _, rest = line.partition(" ")
        rest = rest.strip()
        if k == "v":
            pos.append([F(float(c)) for c in rest.split()[:3]])
        elif k == "f":
            idx = []
            for tok in rest.split():
                vi = int(tok.split("/")[0])
                idx.append(len(pos) + vi if vi < 0 else vi - 1)
            for i in range(1, len(idx) - 1):
                tmp.append((idx[0], idx[i], idx[i + 1]))
        elif k in ("o", "g"):
            if tmp:
                flush()
            cur = rest or "unnamed_object"
        elif k == "usemtl":
            if rest:
                new = mat_map.get(rest, -1)
                if new != mat_id and tmp:
                    flush()
                mat_id = new
        elif k == "mtllib":
            name = None
            for ml in open(os.path.join(os.path.dirname(path), rest)):
                ml = ml.strip()
                if ml.startswith("newmtl"):
                    name = ml.split(None, 1)[1].strip()
                    mat_map[name] = len(mats)
                    mats.append({"name": name, "diffuse": [F(0), F(0), F(0)]})
                elif ml.startswith("Kd") and name is not None:
                    mats[-1]["diffuse"] = [F(float(c)) for c in ml.split()[1:4]]
    if tmp or not models:
        flush()
    return {"positions": pos, "models": models, "materials": mats}


MAT_TYPES = {"lambert": 0, "phong": 1, "blinn-phong": 2, "ggx": 3, "ideal-refraction": 4}


def instantiate(cfg, resolve):
    """-> (materials, prims) in the reference's instance order (description.rs:89-148)."""
    meshes = {m["name"]: m for m in cfg.get("mesh", [])}
    mats_cfg = {}
    for m in cfg.get("material", []):
        mats_cfg.setdefault(m["name"], m)                 # find() returns the first match
    objs = {}
    materials, prims = [], []
    for o in cfg.get("object", []):
        mesh = meshes[o["mesh"]]
        tr = compose(o.get("transform", []))
        emission = np.zeros(3, dtype=np.float32)
        if "name" in o:
            for l in cfg.get("light", []):
                if l["object"] == o["name"]:
                    emission = (v3(l["emission"]) * F(float(l.get("intensity", 1.0)))).astype(np.float32)
                    break
        default = -1
        if "material" in o:
            mc = mats_cfg[o["material"]]
            t = MAT_TYPES[mc["type"]]
            color = v3(mc["albedo"] if t == 0 else mc["reflectance"])
            p = [F(0), F(0), F(0)]
            if t in (1, 2):
                p[0] = F(float(mc["alpha"]))
            elif t == 3:
                p[0], p[1] = F(float(mc["roughness"])), F(float(mc["ior"]))
            elif t == 4:
                p[0], p[1] = F(float(mc["ior"])), F(float(mc.get("absorbtance", 0.0)))
            materials.append({"type": t, "color": color, "emission": emission if t == 0 else np.zeros(3, np.float32), "param": p})
            default = len(materials) - 1
        if mesh["type"] == "obj":
            if mesh["name"] not in objs:
                objs[mesh["name"]] = load_obj(resolve(mesh["path"]))
            f = objs[mesh["name"]]
            base = len(materials)
            if default < 0:
                for om in f["materials"]:
                    materials.append({"type": 0, "color": np.array(om["diffuse"], np.float32), "emission": emission, "param": [F(0)] * 3})
            for m in f["models"]:
                mat = default
                if mat < 0:
                    if m["material_id"] < 0:
                        if not m["tris"]:
                            continue
                        raise ValueError("Specified material is not found in mlt file.")
                    mat = base + m["material_id"]
                for tri in m["tris"]:
                    v = np.concatenate([mat_vec(tr, np.array(f["positions"][i], np.float32)) for i in tri])
                    prims.append({"type": 0, "material": mat, "v": v})
        else:
            if default < 0:
                raise ValueError("Material must be specified")
            c = mat_vec(tr, np.zeros(3, np.float32))
            prims.append({"type": 1, "material": default, "v": np.array([c[0], c[1], c[2], F(float(mesh["radius"]))], np.float32)})
    return materials, prims


def load(path_or_text, asset_root, is_text=False):
    text = path_or_text if is_text else open(path_or_text, "rb").read().decode()
    cfg = tomli.loads(text)

    def resolve(p):
        return p if os.path.isfile(p) else os.path.join(asset_root, p)
    w, h = cfg["film"]["resolution"]
    r = cfg["renderer"]
    out = {
        "renderer": {"samples": r["samples"], "depth": r.get("depth", 5), "depth_limit": r.get("depth-limit", 64),
                     "no_direct_emitter": int(bool(r.get("no-direct-emitter", False))), "threads": r.get("threads", 0),
                     "integrator": {"pt": 0, "pt-direct": 1}[r.get("integrator", "pt-direct")]},
        "film": {"resolution": [w, h], "output": {"png": 0, "hdr": 1}[cfg["film"]["output"]], "gamma": F(float(cfg["film"].get("gamma", 2.2)))},
        "camera": camera(cfg, w, h),
    }
    out["materials"], out["prims"] = instantiate(cfg, resolve)
    return out
