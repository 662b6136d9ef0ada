"""Render rate of one scene with the host SAH tree, the device PLOC tree and the device LBVH (LR_DEVICE_BVH=lbvh), plus the build
times.  usage: bvh_quality_probe.py [scene W H spp]      (one process per builder: the choice is read at lr_scene_create)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "mesh-box.toml"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1370, 256)
if len(sys.argv) > 5 and sys.argv[5] == "--child":
    from lumillyrender_amd import device, host
    mode = sys.argv[6]
    t0 = time.perf_counter()
    d = host.Description(os.path.join(ROOT, "scenes", name)); d.set_resolution(W, H)
    t_host = time.perf_counter() - t0
    sc = device.Scene(d, device_bvh=(mode != "host-sah"))
    tiles, n = host.full_tile(W, H)
    best = 0.0
    for rep in range(3):
        p = d.render_params(spp=spp, seed=rep)
        t0 = time.perf_counter(); sc.render_device(p, tiles, n); dt = time.perf_counter() - t0
        if rep: best = max(best, W * H * spp / dt / 1e6)
    st = sc.stats()
    from lumillyrender_amd import abi
    sc.render_device(d.render_params(spp=min(spp, 8), seed=5, flags=abi.LR_FLAG_COUNT), tiles, n)
    cs = sc.stats()
    counts = {"boxes_per_segment": round(cs.node_visits / max(cs.segments, 1), 2), "prims_per_segment": round(cs.prim_tests / max(cs.segments, 1), 2),
              "boxes_per_shadow_ray": round(cs.shadow_node_visits / max(cs.shadow_rays, 1), 2), "prims_per_shadow_ray": round(cs.shadow_prim_tests / max(cs.shadow_rays, 1), 2)}
    print(json.dumps({"builder": mode, **counts, "scene": name, "Msamples_s": round(best, 1), "device_build_ms": round(st.bvh_build_ms, 3), "scene_create_ms": round(st.upload_ms, 1),
                      "description_load_s (OBJ parse + host SAH build)": round(t_host, 3), "segments": int(st.segments)}))
    sys.exit(0)
for mode, env in (("host-sah", {}), ("device-ploc", {"LR_DEVICE_BVH": "ploc"}), ("device-lbvh", {"LR_DEVICE_BVH": "lbvh"}), ("host-sah", {}), ("device-ploc", {"LR_DEVICE_BVH": "ploc"})):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), name, str(W), str(H), str(spp), "--child", mode], env=dict(os.environ, **env), capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1], flush=True)
