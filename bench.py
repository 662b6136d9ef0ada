#!/usr/bin/env python3
"""bench.py -- the headline measurement: Msamples/s of the per-pixel sampling loop on MI355X.

Workload (BASELINE.json configs[1]): the reference's new-cbox scene (scenes/cbox-spheres.toml here),
1024x1024, 1024 spp, pt-direct (next-event estimation), Lambert only.  One "step" = one full render
of the frame through the C ABI (lr_render: wavefront kernels + film read-back).  The scene (BVH,
primitives, materials, emitters) is resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU, pixel tiles of the frame sharded round-robin over ranks, scene
replicated, no collective on the data path; the film tiles are summed onto rank 0 over gloo after
each render (host gather).  --scaling weak (default) keeps per-GPU work fixed: the frame is rendered at
1024*N spp, each rank owning 1/N of the pixels; --scaling strong keeps the frame at 1024 spp.

The JSON line also carries
  roofline     for the dominant kernel (k_resident, or k_trace when streaming): algorithmic bytes per launch / mean launch duration,
               the duration measured with HIP events around the launches inside the timed region
  cpu_baseline the CPU oracle (a port of the reference algorithm, oracle/) on this box's host cores,
               on a bounded sample of the same frame
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cbox-spheres.toml")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--streaming", action="store_true", help="force the multi-kernel streaming pipeline")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the barrier (nccl = RCCL; testing: gloo)")
    ap.add_argument("--same-device", action="store_true", help="testing on a 1-GPU box: every rank renders on GPU 0")
    return ap.parse_args()


def cpu_baseline(desc, args):
    """Oracle (kind 'port') on all host cores, literal reference traversal (bvh.rs/aabb.rs), on a
    bounded sample: the same 1024x1024 frame at a reduced spp chosen to take ~cpu-seconds."""
    from oracle import binding as oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:        # a cgroup CPU quota (cpu.max = "quota period") caps what the host threads can really use
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except Exception:
        pass
    p = desc.render_params(spp=1, seed=0)
    _, st = oracle.render(desc, p, threads=cores, mode=oracle.BVH, pad=0.0, with_stats=True)
    rate1 = st.samples / max(st.seconds, 1e-9)
    spp = int(max(1, min(64, args.cpu_seconds * rate1 / (args.width * args.height))))
    p = desc.render_params(spp=spp, seed=0)
    _, st = oracle.render(desc, p, threads=cores, mode=oracle.BVH, pad=0.0, with_stats=True)
    return {
        "value": round(st.samples / st.seconds / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"{args.width}x{args.height} frame at {spp} spp ({st.samples} samples, {st.seconds:.1f} s), "
                  "oracle in reference-literal BVH mode, one thread per usable core (affinity and cgroup quota)",
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    from lumillyrender_amd import abi, device, host, multigpu

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
        # host side (film gather, max of the timings): gloo, the default group.  Device side (the barrier that
        # brackets the timed region): an RCCL group when it comes up; a machine where it does not still gets its
        # numbers through the gloo barrier + device synchronisation.
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        host_group = None
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

    W, H = args.width, args.height
    spp = args.spp * (world if args.scaling == "weak" else 1)
    desc = host.Description(os.path.join(ROOT, "scenes", args.scene))
    desc.set_resolution(W, H)
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
        params = desc.render_params(spp=spp, seed=i, integrator=abi.LR_INTEGRATOR_PT_DIRECT, flags=flags, path_slots=args.slots)
        scene.render(params, tiles, n_tiles, out=canvas)      # blocks until the film tiles are on the host
        st = scene.stats()
        shared_film.collect()                                   # every rank's tiles are in the shared film; a barrier publishes it to rank 0
        return st

    for i in range(args.warmup):
        step(1000 + i)
    acc = {"segments": 0, "shadow": 0, "samples": 0, "iterations": 0, "render_ms": 0.0,
           "kernel_ms": [0.0] * abi.LR_K_COUNT, "kernel_timed": [0] * abi.LR_K_COUNT, "kernel_launches": [0] * abi.LR_K_COUNT}
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        st = step(i)
        acc["segments"] += st.segments; acc["shadow"] += st.shadow_rays; acc["samples"] += st.samples
        acc["iterations"] += st.iterations; acc["render_ms"] += st.render_ms
        for k in range(abi.LR_K_COUNT):
            acc["kernel_ms"][k] += st.kernel_ms[k]; acc["kernel_timed"][k] += st.kernel_timed[k]; acc["kernel_launches"][k] += st.kernel_launches[k]
    barrier()
    elapsed = time.perf_counter() - t0
    my_pixels = sum(tiles[i].w * tiles[i].h for i in range(n_tiles))
    assert acc["samples"] == my_pixels * spp * args.steps, f"device finished {acc['samples']} samples, expected {my_pixels * spp * args.steps}"
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_samples = float(W) * H * spp * args.steps
    value = total_samples / elapsed / 1e6

    out = None
    if rank == 0:
        assert np.isfinite(canvas).all(), "non-finite film"
        out = {
            "metric": "Msamples/sec (whole node), new-cbox 1024x1024 pt-direct", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.scene} (the reference's scenes/new-cbox.toml with authored Cornell meshes) "
                            f"{W}x{H} {spp} spp pt-direct, Lambert only",
                "width": W, "height": H, "spp": spp, "integrator": "pt-direct", "tile": args.tile,
                "parallelism": f"pixel tiles round-robin over {world} GPU(s), replicated scene, film assembled in host shared memory",
                "path_slots": args.slots or "library default",
            },
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
            pc = desc.render_params(spp=32, seed=77, integrator=abi.LR_INTEGRATOR_PT_DIRECT, flags=abi.LR_FLAG_COUNT, path_slots=args.slots)
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
            if dom == abi.LR_K_RESIDENT:        # one launch renders the whole frame
                units = float(W) * H * spp
                bytes_per_launch = bytes_per_sample * units
                unit_name = "camera samples"
            else:                               # streaming pipeline: the trace kernel's own share per segment
                units = acc["segments"] / max(acc["kernel_launches"][abi.LR_K_TRACE], 1)
                bytes_per_launch = (32.0 + 8.0 + 4.0 + 32.0 * sc.node_visits / max(sc.segments, 1) + 48.0 * sc.prim_tests / max(sc.segments, 1) / (64.0 if flat else 1.0)) * units
                unit_name = "segments"
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(f"k_{names[dom]}_hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            out["roofline"] = {
                "kernel": f"k_{names[dom]}", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_launch_ms": round(avg_ms, 5), "units_per_launch": round(units, 1), "unit": "GB/s", "units": unit_name,
                "algorithmic_bytes_per_launch": round(bytes_per_launch, 0), "bytes_per_camera_sample": round(bytes_per_sample, 1),
                "boxes_per_query": round(v_per_q, 2), "prim_tests_per_query": round(t_per_q, 2),
                "timed_launches": acc["kernel_timed"][dom],
                # the MEASURED HBM rate next to the algorithmic one (PMC bytes of profiles/traffic.json over this run's launch time)
                "measured_hbm_GBps": round(traffic / (avg_ms * 1e-3) / 1e9, 2) if traffic else None,
                "note": ("the algorithmic bytes are what a wavefront design streams per sample (SURVEY 8d); the resident kernel keeps "
                         "them in LDS / the scalar cache, so frac is not HBM saturation: its real bound is VALU issue "
                         "(PMC: 69 % of the wave64 issue rate, DESIGN.md section 6)") if dom == abi.LR_K_RESIDENT else
                        "streaming pipeline: path state moves through HBM / Infinity Cache every iteration",
            }
            out["kernels_ms_per_launch"] = {names[k]: round(acc["kernel_ms"][k] / acc["kernel_timed"][k], 5)
                                            for k in range(abi.LR_K_COUNT) if acc["kernel_timed"][k]}
            out["path_stats"] = {"segments_per_sample": round(s_per, 3), "shadow_rays_per_sample": round(q_per, 3),
                                 "pipeline": "resident" if dom == abi.LR_K_RESIDENT else "streaming"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, args)
        print(json.dumps(out), flush=True)
    scene.close()
    shared_film.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
