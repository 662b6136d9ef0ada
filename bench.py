#!/usr/bin/env python3
"""bench.py -- the headline measurement: Msamples/s of the per-pixel sampling loop on MI355X.

Default workload = BASELINE.json configs[1] ("c2"): the reference's new-cbox scene (scenes/cbox-spheres.toml here),
1024x1024, 1024 spp, pt-direct (next-event estimation), Lambert only.  One "step" = one full render of the frame
through the C ABI (lr_render: wavefront kernels + read-back of the rendered tiles).  The scene (BVH, primitives,
materials, emitters, sky) is resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps K --warmup W [--config c2|c3|c4|c5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--config selects the other BASELINE.json configs at their full sizes (single-GPU lines of the 8-GPU configs):
  c3 brdf-row.toml 960x540 4096 spp (GGX row)      c4 mesh-box.toml 1920x1370 2048 spp pt (100k-triangle mesh)
  c5 ibl-lens.toml 2048x2048 8192 spp (thin lens, IBL sky, GGX mesh)

Multi-GPU: one process per GPU, pixel tiles of the frame sharded round-robin over ranks, scene replicated, no
collective on the data path; every rank's lr_render writes its tiles into one film in host shared memory.
--scaling strong (default) keeps the frame at the config's spp, so the N-GPU line is the SAME workload as the 1-GPU
line; --scaling weak renders spp*N per pixel (per-GPU work fixed).

The JSON line also carries
  roofline       SURVEY 8(d) block for the dominant kernel: algorithmic bytes per launch / mean launch duration (HIP
                 events around the launches inside the timed region) against the 8 TB/s HBM peak, with the MEASURED
                 HBM bytes (`traffic`, from profiles/<round>_traffic_<config>.json, attached only when the workload
                 recorded there is this run's)
  roofline_valu  the roof the resident kernel is really under: VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU,
                 profiles/<round>_pmc_<config>.json) / launch duration against 1024 SIMDs x clock / 2 cycles per wave64
                 instruction (MI355X_MICROARCH.md: v_fma_f32 2 cycles), transcendentals weighted twice
  cpu_baseline   the CPU oracle (a port of the reference algorithm, oracle/, built -O3 -mavx2) on this box's host cores,
                 on a bounded sample of the same frame
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
VALU_CYCLES_PER_WAVE_INSTR = 2.0   # MI355X_MICROARCH.md cycle constants: v_fma_f32 (wave64) 2 cycles on a SIMD-32

CONFIGS = {
    # name: scene, W, H, spp, integrator (None = the scene file's), description for config.workload, metric text
    "c2": ("cbox-spheres.toml", 1024, 1024, 1024, 1, "the reference's scenes/new-cbox.toml with authored Cornell meshes; pt-direct, Lambert only",
           "Msamples/sec (whole node), new-cbox 1024x1024 pt-direct"),
    "c3": ("brdf-row.toml", 960, 540, 4096, None, "the reference's scenes/brdf.toml; GGX row + Lambert, pt-direct",
           "Msamples/sec (whole node), brdf 960x540 4096 spp"),
    "c4": ("mesh-box.toml", 1920, 1370, 2048, None, "the reference's scenes/sample.toml with a procedural 100k-triangle mesh; pt",
           "Msamples/sec (whole node), sample.toml-class 100k-triangle mesh 1920x1370 2048 spp pt"),
    "c5": ("ibl-lens.toml", 2048, 2048, 8192, None, "the reference's scenes/welcome-2018.toml class: thin lens, HDR IBL sky, GGX mesh; pt-direct",
           "Msamples/sec (whole node), welcome-2018-class 2048x2048 8192 spp IBL"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE.json config (default c2 = configs[1], the headline)")
    ap.add_argument("--scene", default=None, help="override the config's scene file")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of each baseline sample")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--streaming", action="store_true", help="force the multi-kernel streaming pipeline")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the barrier (nccl = RCCL; testing: gloo)")
    ap.add_argument("--same-device", action="store_true", help="testing on a 1-GPU box: every rank renders on GPU 0")
    ap.add_argument("--dump-film", default=None, help="rank 0 saves the last film as .npy (tests)")
    return ap.parse_args()


def usable_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:        # a cgroup CPU quota (cpu.max = "quota period") caps what the host threads can really use
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline(desc, W, H, integ, cpu_seconds):
    """Oracle (kind 'port', the -O3 -mavx2 build) on all usable host cores, on a bounded sample: the same frame at a
    reduced spp chosen to take ~cpu_seconds.  `value` is the reference-literal mode (bvh.rs:131-141 candidate list, one Vec
    per ray); `optimized` is the ordered early-out traversal of BASELINE.md section 3 on the same sample."""
    from oracle import binding as oracle
    cores = usable_cores()

    def timed(mode, pad):
        p = desc.render_params(spp=1, seed=0, integrator=integ)
        _, st = oracle.render(desc, p, threads=cores, mode=mode, pad=pad, with_stats=True, fast=True)
        rate1 = st.samples / max(st.seconds, 1e-9)
        spp = int(max(1, min(64, cpu_seconds * rate1 / (W * H))))
        p = desc.render_params(spp=spp, seed=0, integrator=integ)
        _, st = oracle.render(desc, p, threads=cores, mode=mode, pad=pad, with_stats=True, fast=True)
        return st, spp
    st, spp = timed(oracle.BVH, 0.0)
    st2, spp2 = timed(oracle.BVH_ORDERED, 0.05)
    return {
        "value": round(st.samples / st.seconds / 1e6, 3), "unit": "Msamples/s", "cores": cores,
        "hardware_threads": os.cpu_count() or cores, "kind": "port", "build": "g++ -O3 -mavx2 -ffp-contract=off (oracle/liboracle_fast.so)",
        "sample": f"{W}x{H} frame at {spp} spp ({st.samples} samples, {st.seconds:.1f} s), oracle in reference-literal BVH mode "
                  "(collect every overlapped leaf, then min), one thread per usable core (affinity and cgroup quota)",
        "optimized": {"value": round(st2.samples / st2.seconds / 1e6, 3), "unit": "Msamples/s",
                      "sample": f"{W}x{H} frame at {spp2} spp ({st2.seconds:.1f} s), ordered early-out traversal of the same tree, row tasks"},
    }


def load_profile(kind, cfg, want):
    """Newest (by the `when` stamp the export recorded, then by name) profiles/r*_{kind}_{cfg}.json whose recorded workload
    equals `want` on every key of `want` (scene, film size, spp, slot count); None when there is none -- a figure from
    another workload is never attached."""
    best, best_key = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{kind}_{cfg}.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        wl = d.get("workload", {})
        if all(str(wl.get(k)) == str(v) for k, v in want.items()):
            key = (int(wl.get("when", 0) or 0), os.path.basename(path))
            if best_key is None or key > best_key:
                best, best_key = (path, d), key
    return best


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    from lumillyrender_amd import abi, device, host, multigpu

    scene_file, W, H, base_spp, integ, what, metric = CONFIGS[args.config]
    scene_file = args.scene or scene_file
    W, H = args.width or W, args.height or H
    base_spp = args.spp or base_spp
    stated = (scene_file, W, H, base_spp) == CONFIGS[args.config][:4]

    dist = None
    host_group = None
    dev_group = None
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"     # the env switch exercises the RCCL path on a 1-GPU box
    if use_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.same_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        # host side (film assembly, max of the timings): gloo, the default group.  Device side (the barrier that
        # brackets the timed region): an RCCL group when it comes up; a machine where it does not still gets its
        # numbers through the gloo barrier + device synchronisation.
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        if "nccl" in args.backend:
            try:
                import datetime
                dev_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
                probe = torch.ones(1, device=f"cuda:{local_rank}")
                dist.all_reduce(probe, group=dev_group)
                torch.cuda.synchronize(local_rank)
                assert int(probe.item()) == world
            except Exception as e:                                     # noqa: BLE001 -- report and fall back
                print(f"[bench] rank {rank}: RCCL group unavailable ({type(e).__name__}: {e}); using the gloo barrier", file=sys.stderr, flush=True)
                dev_group = None
    dev_index = local_rank if world > 1 else 0

    spp = base_spp * (world if args.scaling == "weak" else 1)
    desc = host.Description(os.path.join(ROOT, "scenes", scene_file))
    desc.set_resolution(W, H)
    integ_eff = desc.renderer.integrator if integ is None else integ
    scene = device.Scene(desc, device=dev_index)            # scene resident in HBM from here on
    tiles, n_tiles = multigpu.shard_tiles(W, H, args.tile, rank, world)
    flags = (0 if args.no_profile else abi.LR_FLAG_PROFILE) | (abi.LR_FLAG_STREAMING if args.streaming else 0)
    shared_film = multigpu.SharedFilm(W, H, args.tile, dist, dst=0, group=host_group)   # one film in /dev/shm for the ranks of this node
    canvas = shared_film.array
    barrier_buf = torch.zeros(1, device=f"cuda:{dev_index}") if (use_dist and dev_group is not None) else None

    def barrier():
        torch.cuda.synchronize(dev_index)
        if dist is not None:
            if dev_group is not None:
                dist.all_reduce(barrier_buf, group=dev_group)           # RCCL over xGMI
            else:
                dist.barrier()
        torch.cuda.synchronize(dev_index)

    def step(i):
        params = desc.render_params(spp=spp, seed=i, integrator=integ, flags=flags, path_slots=args.slots)
        scene.render(params, tiles, n_tiles, out=canvas)      # blocks until this rank's tiles are in the (shared) host film
        st = scene.stats()
        shared_film.collect()                                   # every rank's tiles are in the shared film; a barrier publishes it to rank 0
        return st

    for i in range(args.warmup):
        step(1000 + i)
    acc = {"segments": 0, "shadow": 0, "samples": 0, "iterations": 0, "render_ms": 0.0,
           "kernel_ms": [0.0] * abi.LR_K_COUNT, "kernel_timed": [0] * abi.LR_K_COUNT, "kernel_launches": [0] * abi.LR_K_COUNT}
    path_slots = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        st = step(i)
        acc["segments"] += st.segments; acc["shadow"] += st.shadow_rays; acc["samples"] += st.samples
        acc["iterations"] += st.iterations; acc["render_ms"] += st.render_ms
        path_slots = int(st.path_slots)
        for k in range(abi.LR_K_COUNT):
            acc["kernel_ms"][k] += st.kernel_ms[k]; acc["kernel_timed"][k] += st.kernel_timed[k]; acc["kernel_launches"][k] += st.kernel_launches[k]
    barrier()
    elapsed = time.perf_counter() - t0
    my_pixels = sum(tiles[i].w * tiles[i].h for i in range(n_tiles))
    assert acc["samples"] == my_pixels * spp * args.steps, f"device finished {acc['samples']} samples, expected {my_pixels * spp * args.steps}"
    rank_ms = [acc["render_ms"] / max(args.steps, 1)]
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        all_ms = [None] * world
        dist.all_gather_object(all_ms, rank_ms[0])
        rank_ms = all_ms

    total_samples = float(W) * H * spp * args.steps
    value = total_samples / elapsed / 1e6

    out = None
    if rank == 0:
        assert np.isfinite(canvas).all(), "non-finite film"
        if args.dump_film:
            np.save(args.dump_film, np.array(canvas))
        integ_name = "pt-direct" if integ_eff == abi.LR_INTEGRATOR_PT_DIRECT else "pt"
        out = {
            "metric": metric if stated else f"Msamples/sec (whole node), {scene_file} {W}x{H}", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {scene_file} ({what}) {W}x{H} {spp} spp {integ_name}" + ("" if stated else " [NOT the stated config: overridden on the command line]"),
                "baseline_config": args.config, "width": W, "height": H, "spp": spp, "integrator": integ_name, "tile": args.tile,
                "parallelism": f"pixel tiles round-robin over {world} GPU(s), replicated scene, film assembled in host shared memory, no collective on the data path",
                "path_slots": path_slots,
            },
            "rank_render_ms": {"max": round(max(rank_ms), 3), "min": round(min(rank_ms), 3)},
        }
        # ---- roofline of the dominant kernel, N = 1 only ---------------------------------------------
        names = abi.LR_KERNEL_NAMES
        dom = max(range(abi.LR_K_COUNT), key=lambda k: acc["kernel_ms"][k] / max(acc["kernel_timed"][k], 1) * acc["kernel_launches"][k])
        if world == 1 and acc["kernel_timed"][dom] > 0:
            # DESIGN.md "algorithmic bytes" = SURVEY 8(d) with this build's record sizes, per camera sample:
            #   224 B per closest-hit segment (ray 32 r + hit/list 16 w in trace; 96 r + 80 w of ray, hit, throughput,
            #   radiance in shade), 108 B per shadow ray (48 w + 48 r + 12 rw), 32 B per child box tested, 48 B per
            #   primitive tested (1/64 of that when the primitive rows come through the scalar cache once per wave),
            #   12 B per film pixel.  Box / primitive counts come from one short counted render (LR_FLAG_COUNT).
            pc = desc.render_params(spp=min(32, spp), seed=77, integrator=integ, flags=abi.LR_FLAG_COUNT, path_slots=args.slots)
            scene.render(pc, tiles, n_tiles, out=canvas)
            sc = scene.stats()
            flat = desc.desc.n_prims <= 32
            queries = max(sc.segments + sc.shadow_rays, 1)
            v_per_q = (sc.node_visits + sc.shadow_node_visits) / queries
            t_per_q = (sc.prim_tests + sc.shadow_prim_tests) / queries
            s_per = acc["segments"] / acc["samples"]
            q_per = acc["shadow"] / acc["samples"]
            scene_bytes_per_q = 32.0 * v_per_q + 48.0 * t_per_q / (64.0 if flat else 1.0)
            bytes_per_sample = 224.0 * s_per + 108.0 * q_per + scene_bytes_per_q * (s_per + q_per) + 12.0 / spp
            avg_ms = acc["kernel_ms"][dom] / acc["kernel_timed"][dom]
            resident = dom == abi.LR_K_RESIDENT
            if resident:                        # one launch renders the whole frame
                units = float(W) * H * spp
                bytes_per_launch = bytes_per_sample * units
                unit_name = "camera samples"
            else:                               # streaming pipeline: the dominant kernel's own share per entry it processes
                if dom == abi.LR_K_SHADOW:
                    units = acc["shadow"] / max(acc["kernel_launches"][dom], 1)
                    per_unit = 48.0 + 12.0 + 4.0 + 32.0 * sc.shadow_node_visits / max(sc.shadow_rays, 1) + 48.0 * sc.shadow_prim_tests / max(sc.shadow_rays, 1) / (64.0 if flat else 1.0)
                    unit_name = "shadow rays"
                elif dom == abi.LR_K_SHADE:
                    units = (acc["segments"] + acc["samples"] * 0) / max(acc["kernel_launches"][dom], 1)
                    per_unit = 96.0 + 80.0 + 4.0 + 64.0
                    unit_name = "path vertices"
                else:
                    units = acc["segments"] / max(acc["kernel_launches"][abi.LR_K_TRACE], 1)
                    per_unit = 32.0 + 8.0 + 4.0 + 32.0 * sc.node_visits / max(sc.segments, 1) + 48.0 * sc.prim_tests / max(sc.segments, 1) / (64.0 if flat else 1.0)
                    unit_name = "segments"
                bytes_per_launch = per_unit * units
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            want = {"scene": scene_file, "width": W, "height": H}
            want.update({"spp": spp} if resident else {"path_slots": path_slots})
            traffic, traffic_src = None, None
            got = load_profile("traffic", args.config, want)
            if got:
                traffic = got[1].get(f"k_{names[dom]}_hbm_bytes_per_launch")
                traffic_src = os.path.relpath(got[0], ROOT)
            out["roofline"] = {
                "kernel": f"k_{names[dom]}", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_ms, 5), "units_per_launch": round(units, 1), "units": unit_name,
                "algorithmic_bytes_per_launch": round(bytes_per_launch, 0), "bytes_per_camera_sample": round(bytes_per_sample, 1),
                "boxes_per_query": round(v_per_q, 2), "prim_tests_per_query": round(t_per_q, 2),
                "timed_launches": acc["kernel_timed"][dom],
                # the MEASURED HBM rate next to the algorithmic one (PMC bytes of a profile of THIS workload over this run's launch time)
                "measured_hbm_GBps": round(traffic / (avg_ms * 1e-3) / 1e9, 2) if traffic else None,
                "measured_frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                "nominal": bool(resident),
                "note": ("NOMINAL: the algorithmic bytes are what a wavefront design streams per sample (SURVEY 8d); the resident kernel keeps "
                         "them in LDS / the scalar cache and moves only `traffic` bytes through HBM, so this frac is not HBM saturation -- "
                         "the roof this kernel is under is VALU issue, see roofline_valu") if resident else
                        "streaming pipeline: path state moves through HBM / Infinity Cache every iteration; traffic = PMC bytes of this kernel per launch",
            }
            # ---- streaming pipeline: the bandwidth-bound stage beside the traversal (k_shade_all), MEASURED bytes over this run's launch time ----
            if not resident and got and acc["kernel_timed"][abi.LR_K_SHADE]:
                sh_bytes = got[1].get("k_shade_all_hbm_bytes_per_launch")
                if sh_bytes:
                    sh_ms = acc["kernel_ms"][abi.LR_K_SHADE] / acc["kernel_timed"][abi.LR_K_SHADE]
                    alone = got[1].get("k_shade_all_hbm_GBps_in_pmc_pass")
                    out["roofline_shade"] = {
                        "kernel": "k_shade_all", "bound": "hbm", "achieved": round(sh_bytes / (sh_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(sh_bytes / (sh_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": sh_bytes, "traffic_source": traffic_src,
                        "avg_launch_ms": round(sh_ms, 5), "alone_GBps_in_pmc_pass": alone,
                        "alone_frac": round(alone / HBM_PEAK_GBS, 5) if alone else None,
                        "note": "PMC bytes (FETCH_SIZE x2 + WRITE_SIZE) of the committed pass of this workload over this run's launch time; the launch "
                                "shares the GPU with the other slot groups' traversal kernels -- alone (serialised profiler pass) it moves alone_GBps",
                    }
            # ---- the VALU-issue roof (what bounds k_resident): instructions from a committed PMC profile of this workload ----
            gotp = load_profile("pmc", args.config, want)
            if gotp:
                kk = [k for k in gotp[1].get("kernels", {}) if k.split("<")[0] == f"k_{names[dom]}"]
                if kk:
                    e = gotp[1]["kernels"][kk[0]]
                    valu, trans = e.get("SQ_INSTS_VALU"), e.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
                    gui, us = e.get("GRBM_GUI_ACTIVE"), e.get("avg_us_in_pmc_pass")
                    if valu and gui and us:
                        clock_ghz = gui / 8.0 / us / 1e3                  # GRBM_GUI_ACTIVE sums the 8 XCDs (MI355X_MICROARCH.md, DVFS section)
                        weighted = valu + trans                            # a transcendental occupies the issue port twice as long
                        peak = N_SIMD * clock_ghz / VALU_CYCLES_PER_WAVE_INSTR          # G wave-instructions / s
                        ach = weighted / (avg_ms * 1e-3) / 1e9
                        wc = e.get("SQ_WAVE_CYCLES")
                        out["roofline_valu"] = {
                            "kernel": kk[0], "bound": "valu-issue", "achieved": round(ach, 2), "peak": round(peak, 2), "unit": "G wave-instr/s",
                            "frac": round(ach / peak, 4), "valu_wave_instr_per_launch": valu, "transcendental_per_launch": trans,
                            "clock_GHz_in_pmc_pass": round(clock_ghz, 3), "source": os.path.relpath(gotp[0], ROOT),
                            "lanes_per_valu_instr": round(e["SQ_THREAD_CYCLES_VALU"] / valu, 1) if e.get("SQ_THREAD_CYCLES_VALU") else None,
                            "wave_cycles_parked": round(e["SQ_WAIT_ANY"] / wc, 3) if wc and e.get("SQ_WAIT_ANY") else None,
                            "wave_cycles_issue_stalled": round(e["SQ_WAIT_INST_ANY"] / wc, 3) if wc and e.get("SQ_WAIT_INST_ANY") else None,
                            "note": "instructions per launch from the committed rocprofv3 pass of the same workload (deterministic per scene, "
                                    "film, spp up to the seed), duration from this run's HIP events; peak = 1024 SIMDs x clock / 2 cycles",
                        }
            out["kernels_ms_per_launch"] = {names[k]: round(acc["kernel_ms"][k] / acc["kernel_timed"][k], 5)
                                            for k in range(abi.LR_K_COUNT) if acc["kernel_timed"][k]}
            out["path_stats"] = {"segments_per_sample": round(s_per, 3), "shadow_rays_per_sample": round(q_per, 3),
                                 "pipeline": "resident" if resident else "streaming"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, W, H, integ, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    scene.close()
    shared_film.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
