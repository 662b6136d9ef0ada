"""Random small scenes, device film vs oracle film.  usage: fuzz_parity.py first_seed n_seeds [W H spp [max_objects [hostile]]]
A scene: 3..60 primitives (spheres and transformed quads, so both the flat loop and the 4-wide tree are used), all five
BSDFs with random parameters, one or two area lights or a bright sky, pinhole / thin-lens / omnidirectional camera,
pt or pt-direct.  Reports the worst |device - oracle| relative to max(1, |oracle|) per scene."""
import sys
import numpy as np
sys.path.insert(0, ".")


def scene_text(seed, W, H, max_objs=26, hostile=False):
    """hostile=True draws parameters from the corners where the reference's arithmetic produces infinities and NaNs
    (non-integral Phong exponents, mirror-like or rough-as-chalk GGX, ior 1, black and over-unit albedos, zero-area and
    needle-thin quads, pin-head and planet-size spheres): the device must reproduce the same NaN mask and, where the
    oracle is finite, the same values."""
    r = np.random.default_rng(seed)
    u = lambda a, b: float(r.uniform(a, b))
    v3 = lambda a, b: "[%.6g, %.6g, %.6g]" % (u(a, b), u(a, b), u(a, b))
    pick = lambda xs: xs[int(r.integers(0, len(xs)))]
    mats, names = [], []
    n_mat = int(r.integers(2, 7))
    for i in range(n_mat):
        kind = ["lambert", "phong", "blinn-phong", "ggx", "ideal-refraction"][int(r.integers(0, 5))] if i else "lambert"
        name = f"m{i}"
        if hostile and r.random() < 0.6:
            # (no albedo above 1: an energy-gaining surface makes the throughput overflow after ~80 bounces in the
            #  throughput form but not in the reference's inside-out recursion -- different garbage on both sides)
            col = pick(["[0, 0, 0]", "[1, 1, 1]", "[1, 0, 1]", "[1e-30, 1e-30, 1e-30]", v3(0, 1)])
            if kind == "lambert":
                mats.append(f'[[material]]\nname = "{name}"\ntype = "lambert"\nalbedo = {col}')
            elif kind in ("phong", "blinn-phong"):
                mats.append(f'[[material]]\nname = "{name}"\ntype = "{kind}"\nreflectance = {col}\nalpha = {pick(["0", "0.5", "1.5", "2.75", "100", "1000", "1e6", "-1"])}')
            elif kind == "ggx":
                mats.append(f'[[material]]\nname = "{name}"\ntype = "ggx"\nreflectance = {col}\nroughness = {pick(["0", "1e-4", "0.01", "1", "3"])}\nior = {pick(["1", "1.0001", "1e5", "0", "0.5"])}')
            else:
                mats.append(f'[[material]]\nname = "{name}"\ntype = "ideal-refraction"\nreflectance = {col}\nior = {pick(["1", "1.0001", "0.5", "10", "1e5"])}\nabsorbtance = {pick(["0", "1", "100"])}')
        elif kind == "lambert":
            mats.append(f'[[material]]\nname = "{name}"\ntype = "lambert"\nalbedo = {v3(0.1, 0.95)}')
        elif kind in ("phong", "blinn-phong"):
            mats.append(f'[[material]]\nname = "{name}"\ntype = "{kind}"\nreflectance = {v3(0.2, 0.95)}\nalpha = {int(r.integers(1, 40))}')   # integral: powf(negative cosine, alpha) is NaN otherwise, in the reference too
        elif kind == "ggx":
            mats.append(f'[[material]]\nname = "{name}"\ntype = "ggx"\nreflectance = {v3(0.3, 1.0)}\nroughness = {u(0.15, 0.9):.6g}\nior = {u(1.2, 50):.6g}')
        else:
            mats.append(f'[[material]]\nname = "{name}"\ntype = "ideal-refraction"\nreflectance = {v3(0.6, 1.0)}\nior = {u(1.1, 2.0):.6g}\nabsorbtance = {u(0, 0.02):.6g}')
        names.append(name)
    n_prim_objs = int(r.integers(2, max_objs))
    objs = []
    size = 100.0
    for i in range(n_prim_objs):
        m = names[int(r.integers(0, n_mat))]
        if r.random() < 0.5:
            objs.append(f'[[object]]\nmesh = "ball{i % 3}"\nmaterial = "{m}"\ntransform = [ {{ type = "translate", vector = {v3(-size, size)} }} ]')
        else:
            sx, sz = (u(10, 80), u(10, 80)) if not (hostile and r.random() < 0.25) else (pick([0.0, 1e-4, 300.0]), pick([0.0, 1e-3, 50.0]))
            objs.append(f'[[object]]\nmesh = "panel"\nmaterial = "{m}"\ntransform = [ {{ type = "scale", vector = [{sx:.6g}, 1, {sz:.6g}] }}, '
                        f'{{ type = "axis-angle", axis = {v3(-1, 1)}, angle = {u(0, 360):.6g} }}, {{ type = "translate", vector = {v3(-size, size)} }} ]')
    # floor
    objs.append(f'[[object]]\nmesh = "panel"\nmaterial = "m0"\ntransform = [ {{ type = "scale", vector = [400, 1, 400] }}, {{ type = "translate", vector = [0, {-size - 5:.6g}, 0] }} ]')
    lights = []
    n_light = int(r.integers(0, 3))
    for i in range(n_light):
        objs.append(f'[[object]]\nname = "lamp{i}"\nmesh = "panel"\ntransform = [ {{ type = "axis-angle", axis = [1, 0, 0], angle = 180 }}, '
                    f'{{ type = "scale", vector = [{u(10, 40):.6g}, {u(10, 40):.6g}, {u(10, 40):.6g}] }}, {{ type = "translate", vector = [{u(-60, 60):.6g}, {size + u(0, 40):.6g}, {u(-60, 60):.6g}] }} ]')
        lights.append(f'{{ type = "area", object = "lamp{i}", emission = {v3(3, 20)} }}')
    sky = v3(0.3, 1.5) if n_light == 0 or r.random() < 0.3 else "[0, 0, 0]"
    cam_kind = ["ideal-pinhole", "thin-lens", "omnidirectional"][int(r.integers(0, 3))]
    cam = f'type = "{cam_kind}"\n'
    if cam_kind == "ideal-pinhole":
        cam += f"fov = {u(30, 80):.6g}\n"
    elif cam_kind == "thin-lens":
        cam += f"fov = {u(30, 70):.6g}\nfocus-distance = {u(150, 400):.6g}\nf-number = {u(1.4, 11):.6g}\n"
    cam += f'transform = [ {{ type = "look-at", origin = [{u(-80, 80):.6g}, {u(-20, 90):.6g}, {-u(250, 380):.6g}], target = [{u(-20, 20):.6g}, {u(-20, 20):.6g}, 0], up = [0, 1, 0] }} ]'
    integ = "pt-direct" if (n_light and r.random() < 0.7) else "pt"
    text = f'''mesh = [ {{ name = "panel", type = "obj", path = "models/simple/quad.obj" }}, {{ name = "ball0", type = "sphere", radius = {(pick([1e-3, 0.05, 400.0]) if hostile and r.random() < 0.3 else u(5, 40)):.6g} }}, {{ name = "ball1", type = "sphere", radius = {u(5, 40):.6g} }}, {{ name = "ball2", type = "sphere", radius = {u(2, 15):.6g} }} ]
light = [ {", ".join(lights)} ]

[renderer]
integrator = "{integ}"
samples = 8

[film]
output = "hdr"
resolution = [{W}, {H}]

[sky]
type = "uniform"
color = {sky}

[camera]
{cam}

''' + "\n".join(mats) + "\n\n" + "\n".join(objs) + "\n"
    return text, integ, cam_kind


def run(seed, W, H, spp, max_objs=26, hostile=False):
    from lumillyrender_amd import host, device
    from oracle import binding as oracle
    text, integ, cam = scene_text(seed, W, H, max_objs, hostile)
    desc = host.Description(text=text)
    desc.set_resolution(W, H)
    params = desc.render_params(spp=spp, seed=seed)
    want = oracle.render(desc, params, threads=0)
    scene = device.Scene(desc)
    worst = 0.0
    for flags in (0, 4, 8, 16):                            # default pipeline, forced streaming, forced resident (falls back to the default choice when it does not fit), forced fused
        params.flags = flags
        got = scene.render(params)
        assert scene.stats().samples == W * H * spp
        if hostile:
            # where the reference's own arithmetic leaves the floats (inf * 0, 0 / 0, powf of a negative base) the sample is
            # lost on both sides; WHICH non-finite value comes out depends on the association of the radiance sum (the
            # recursion multiplies a factor by the whole deeper sum, the throughput form by each term: inf * (0 + x) = inf
            # but inf * 0 + inf * x = NaN), so the hostile mode compares the non-finite MASKS and the finite values
            assert np.array_equal(np.isfinite(got), np.isfinite(want)), "non-finite masks differ"
        else:
            assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN masks differ"
        fin = np.isfinite(want) & np.isfinite(got)
        err = float(np.max(np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin])))) if fin.any() else 0.0
        worst = max(worst, err)
    n_prims = desc.desc.n_prims
    scene.close()
    return worst, n_prims, integ, cam, float(np.nanmean(want)), float(np.isnan(want).mean())


if __name__ == "__main__":
    first, n = int(sys.argv[1]), int(sys.argv[2])
    W, H, spp = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (48, 32, 8)
    max_objs = int(sys.argv[6]) if len(sys.argv) > 6 else 26
    hostile = len(sys.argv) > 7 and sys.argv[7] == "hostile"
    bad = 0
    for seed in range(first, first + n):
        try:
            worst, n_prims, integ, cam, mean, lit = run(seed, W, H, spp, max_objs, hostile)
        except Exception as e:                             # a scene the loader rejects is a generator problem, report and go on
            print(f"seed {seed}: ERROR {e}")
            bad += 1
            continue
        flag = "" if worst < 1e-4 else "   <-- ABOVE 1e-4"
        if worst >= 1e-4:
            bad += 1
        print(f"seed {seed}: {n_prims:3d} prims {integ:9s} {cam:15s} worst rel err {worst:.3e}  film mean {mean:.3g} nan {lit:.3f}{flag}")
    print("failures:", bad)
    sys.exit(1 if bad else 0)
