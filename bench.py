#!/usr/bin/env python3
"""bench.py -- the headline measurement: Msamples/s of the per-pixel sampling loop on MI355X.

Default workload = BASELINE.json configs[1] ("c2"): the reference's new-cbox scene (scenes/cbox-spheres.toml here),
1024x1024, 1024 spp, pt-direct (next-event estimation), Lambert only.  One "step" = one full render of the frame
through the C ABI (lr_render: the path kernels + read-back of the rendered tiles).  The scene (BVH, primitives,
materials, emitters, sky) is resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps K --warmup W [--config c2|c3|c4|c5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--config selects the other BASELINE.json configs at their full sizes (single-GPU lines of the 8-GPU configs):
  c3 brdf-row.toml 960x540 4096 spp (GGX row)      c4 mesh-box.toml 1920x1370 2048 spp pt (100k-triangle mesh)
  c5 ibl-lens.toml 2048x2048 8192 spp (thin lens, IBL sky, GGX mesh)
A default run (c2, one GPU) also renders legs of c3, c4 and c5 -- each at its STATED size, c5 with all 8192 spp -- AFTER the headline's
timed region and attaches them as `other_configs`, each with its own `roofline`, `cpu_baseline` and `setup` block, so that one
driver-run line carries a number for every config (about a minute in all).

Multi-GPU: one process per GPU, pixel tiles of the frame dealt diagonally over ranks (lr_host_tiles), scene replicated, no
collective on the data path; every rank's lr_render writes its tiles into one film in host shared memory.
--scaling strong (default) keeps the frame at the config's spp, so the N-GPU line is the SAME workload as the 1-GPU
line; --scaling weak renders spp*N per pixel (per-GPU work fixed).

The JSON line also carries
  roofline       the roof the dominant kernel is under; read `lane_weighted_frac` first (issue-slot utilisation x active lanes per
                 instruction / 64: what share of the chip's VALU lane-slots did work), `frac` is issue-slot utilisation alone.  The path kernels (k_path_flat / k_path_tree / k_resident) keep
                 the path state in registers or LDS and the scene in caches: they are bound by VALU ISSUE, not by HBM, so
                 `bound` = "valu-issue": VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU, transcendentals once
                 more: they hold the port twice as long; from the committed profiles/<round>_pmc_<config>.json of THIS
                 workload) / this run's mean launch duration (HIP events on the launch stream) against 1024 SIMDs x clock / 2
                 cycles per wave64 instruction (MI355X_MICROARCH.md).  `traffic` = measured HBM bytes per launch (FETCH_SIZE x 2 +
                 WRITE_SIZE passes, profiles/<round>_traffic_<config>.json).  A streaming-pipeline kernel (only with
                 --streaming) reports its MEASURED bytes over its launch time against the 8 TB/s HBM peak.
  roofline_hbm_nominal   SURVEY 8(d)'s algorithmic bytes per launch over the launch time -- what a wavefront design would stream;
                 these kernels do not move them (see `traffic`), it is carried for comparison only and flagged `cache_served`
                 when it exceeds what HBM could deliver
  cpu_baseline   the CPU oracle (a port of the reference algorithm, oracle/, built -O3 -mavx2) on this box's host cores,
                 on a bounded sample of the same frame
  setup          scene load, host SAH build seconds (description.rs:67-73's "bvh construction"), lr_scene_create, upload ms, the
                 4-wide tree's node / no-culling counts -- what main.rs:139-144 lumps into `elapse`; never part of `value`
  env_overrides  every LR_* variable set in the environment; the ones that select another library build (PRODUCT_ENV) are REFUSED
                 unless --allow-overrides (the product library itself reads no LR_* switch: csrc/lr_knobs.h)
  ranks          every rank's share of a frame as per-rank lists: device ms, dominant-kernel ms, read-back ms, barrier wait, rays, host
                 BVH build -- what makes a multi-GPU run that falls short diagnosable
  other_configs  c1 (BASELINE configs[0]: the CPU plumbing case -- the oracle's rate on the whole job, the GPU frame only checked
                 against it), c3 / c3p / c3b (GGX / Phong / Blinn-Phong rows), c4, c5
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
NOMINAL_CLOCK_GHZ = 2.4        # used only when the PMC pass recorded no GRBM_GUI_ACTIVE

# device record sizes the SURVEY 8(d) byte model prices (csrc/lr_device.h)
NODE_BYTES_PER_CHILD_BOX = 16.0    # 64-B 4-wide node / 4 child boxes
PRIM_BYTES = 48.0

CONFIGS = {
    # name: scene, W, H, spp, integrator (None = the scene file's), description for config.workload, metric text
    "c1": ("cbox-spheres.toml", 256, 256, 16, 0, "BASELINE configs[0]: the reference's scenes/new-cbox.toml 256x256 16 spp pt -- its CPU-runnable plumbing case",
           "Msamples/sec, new-cbox 256x256 16 spp pt (CPU plumbing case)"),
    "c3p": ("brdf-row-phong.toml", 960, 540, 4096, None, "the reference's scenes/brdf.toml with the GGX row replaced by Phong lobes, alpha in {1, 5, 10, 20} (phong.rs:37-68); pt-direct",
            "Msamples/sec (whole node), brdf 960x540 4096 spp, Phong row"),
    "c3b": ("brdf-row-blinn-phong.toml", 960, 540, 4096, None, "the reference's scenes/brdf.toml with the GGX row replaced by Blinn-Phong lobes, alpha in {1, 5, 10, 20} (blinn_phong.rs:37-72); pt-direct",
            "Msamples/sec (whole node), brdf 960x540 4096 spp, Blinn-Phong row"),
    "c2": ("cbox-spheres.toml", 1024, 1024, 1024, 1, "the reference's scenes/new-cbox.toml with authored Cornell meshes; pt-direct, Lambert only",
           "Msamples/sec (whole node), new-cbox 1024x1024 pt-direct"),
    "c3": ("brdf-row.toml", 960, 540, 4096, None, "the reference's scenes/brdf.toml; GGX row + Lambert, pt-direct",
           "Msamples/sec (whole node), brdf 960x540 4096 spp"),
    "c4": ("mesh-box.toml", 1920, 1370, 2048, None, "the reference's scenes/sample.toml with a procedural 100k-triangle mesh; pt",
           "Msamples/sec (whole node), sample.toml-class 100k-triangle mesh 1920x1370 2048 spp pt"),
    "c5": ("ibl-lens.toml", 2048, 2048, 8192, None, "the reference's scenes/welcome-2018.toml class: thin lens, HDR IBL sky, GGX mesh; pt-direct",
           "Msamples/sec (whole node), welcome-2018-class 2048x2048 8192 spp IBL"),
}
CPU_ONLY_CONFIGS = ("c1",)           # BASELINE configs[0] is the reference's CPU plumbing case: no GPU value, no counter profile (c1_leg)
# legs attached to a default run: (config, steps, warmup, spp or 0 = the stated spp).  Every leg runs at its STATED size, warm-up
# frame included (buffers sized by the frame -- the chunk sums of a band, up to 2 x 1 GiB -- are allocated in the warm-up frame, not in the timed one)
OTHER_LEGS = (("c3", 2, 1, 0), ("c3p", 2, 1, 0), ("c3b", 2, 1, 0), ("c4", 2, 1, 0), ("c5", 1, 1, 0))
# LR_* environment variables that change WHAT is measured: only the library paths are left -- the product library reads no LR_*
# switch since round 6 (csrc/lr_knobs.h: the diagnostic knobs exist in the knob build only, which LR_HIP_LIB would have to select).
# They are recorded in the JSON line and refused unless --allow-overrides.
PRODUCT_ENV = ("LR_HIP_LIB", "LR_HOST_LIB", "LR_ORACLE_LIB")      # (LR_ORACLE_LIB replaces the library the cpu_baseline leg times: oracle/binding.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE.json config (default c2 = configs[1], the headline; c3p / c3b = the Phong / Blinn-Phong rows of configs[2]; "
                    "c1 = configs[0], the CPU plumbing case)")
    ap.add_argument("--scene", default=None, help="override the config's scene file")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0, help="tile edge in pixels (0: lr_host_default_tile)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the c3 / c4 / c5 legs a default run attaches")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of each baseline sample")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--streaming", action="store_true", help="force the multi-kernel streaming pipeline")
    ap.add_argument("--resident", action="store_true", help="force the resident pipeline (path state in LDS)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the barrier (nccl = RCCL; testing: gloo)")
    ap.add_argument("--same-device", action="store_true", help="testing on a 1-GPU box: every rank renders on GPU 0")
    ap.add_argument("--dump-film", default=None, help="rank 0 saves the last film as .npy (tests)")
    ap.add_argument("--allow-overrides", action="store_true", help="run although LR_* variables that change the product path are set (they are recorded in env_overrides)")
    ap.add_argument("--leg-cpu-seconds", type=float, default=5.0, help="target CPU time of each baseline sample of the other_configs legs")
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
    st2, spp2 = timed(oracle.OWNBOX_ORDERED, 0.05)
    return {
        "value": round(st.samples / st.seconds / 1e6, 3), "unit": "Msamples/s", "cores": cores,
        "hardware_threads": os.cpu_count() or cores, "kind": "port", "build": "g++ -O3 -mavx2 -ffp-contract=off (%s)" % os.path.relpath(oracle._FAST_PATH, ROOT),
        "sample": f"{W}x{H} frame at {spp} spp ({st.samples} samples, {st.seconds:.1f} s), oracle in reference-literal BVH mode "
                  "(collect every overlapped leaf, then min), one thread per usable core (affinity and cgroup quota)",
        "optimized": {"value": round(st2.samples / st2.seconds / 1e6, 3), "unit": "Msamples/s",
                      "sample": f"{W}x{H} frame at {spp2} spp ({st2.seconds:.1f} s), ordered early-out traversal of the same tree with the leaf's own-box test (same film bits), row tasks"},
    }


LIB_BUILD = [None]          # device.build_id() of the library being timed (set in main)


def load_profile(kind, cfg, want):
    """Newest (by the `when` stamp the export recorded, then by name) profiles/r*_{kind}_{cfg}*.json whose recorded workload
    equals `want` on every key of `want` (scene, film size, spp, pipeline); None when there is none -- a figure from
    another workload is never attached."""
    best, best_key = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{kind}_{cfg}*.json"))):
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


def new_acc(abi):
    return {"segments": 0, "shadow": 0, "samples": 0, "iterations": 0, "render_ms": 0.0, "path_slots": 0, "pipeline": 0,
            "kernel_ms": [0.0] * abi.LR_K_COUNT, "kernel_timed": [0] * abi.LR_K_COUNT, "kernel_launches": [0] * abi.LR_K_COUNT}


def add_stats(acc, st, abi):
    acc["segments"] += st.segments; acc["shadow"] += st.shadow_rays; acc["samples"] += st.samples
    acc["iterations"] += st.iterations; acc["render_ms"] += st.render_ms
    acc["path_slots"] = int(st.path_slots); acc["pipeline"] = int(st.pipeline)
    for k in range(abi.LR_K_COUNT):
        acc["kernel_ms"][k] += st.kernel_ms[k]; acc["kernel_timed"][k] += st.kernel_timed[k]; acc["kernel_launches"][k] += st.kernel_launches[k]


PIPELINE_NAMES = {0: "streaming", 1: "resident", 2: "fused"}


def roofline_blocks(abi, cfg, scene_file, desc, scene, acc, W, H, spp, integ, tiles, n_tiles, canvas, slots, flags):
    """`roofline` (+ `roofline_hbm_nominal`, `kernels_ms_per_launch`, `path_stats`) for the dominant kernel of the timed steps."""
    out = {}
    names = abi.LR_KERNEL_NAMES
    dom = max(range(abi.LR_K_COUNT), key=lambda k: acc["kernel_ms"][k] / max(acc["kernel_timed"][k], 1) * acc["kernel_launches"][k])
    if acc["kernel_timed"][dom] == 0:
        return out
    avg_ms = acc["kernel_ms"][dom] / acc["kernel_timed"][dom]
    one_launch = dom in (abi.LR_K_RESIDENT, abi.LR_K_PATH)          # one launch renders the whole frame, state never leaves the CU
    pipeline = PIPELINE_NAMES.get(acc["pipeline"], "?")
    s_per = acc["segments"] / max(acc["samples"], 1)
    q_per = acc["shadow"] / max(acc["samples"], 1)
    want = {"scene": scene_file, "width": W, "height": H, "spp": spp}
    scale, scaled_from = 1.0, None
    if load_profile("pmc", cfg, want) is None:
        # no counter pass at this spp: one of the SAME scene and film at another spp, counts scaled by the spp ratio (camera samples are
        # i.i.d. and every per-launch count is linear in their number) -- labelled as such
        loose = {"scene": scene_file, "width": W, "height": H}
        alt = load_profile("pmc", cfg, loose)
        if alt and alt[1].get("workload", {}).get("spp"):
            scaled_from = int(alt[1]["workload"]["spp"])
            scale = spp / float(scaled_from)
            want = dict(loose, spp=scaled_from)

    def kernel_entry(d, prefix):
        for k, v in d.items():
            if k.split("<")[0].startswith(prefix):
                return k, v
        return None, None

    # ---- measured HBM bytes of the dominant kernel (committed FETCH_SIZE / WRITE_SIZE passes of this workload) ----
    traffic, traffic_src = None, None
    got = load_profile("traffic", cfg, want)
    if got:
        for k, v in got[1].items():
            if k.startswith(f"k_{names[dom]}") and k.endswith("_hbm_bytes_per_launch"):
                traffic, traffic_src = v * scale, os.path.relpath(got[0], ROOT)
    measured_gbps = traffic / (avg_ms * 1e-3) / 1e9 if traffic else None

    # ---- the VALU-issue roof, from the committed SQ counter passes of this workload ----
    valu_block = None
    gotp = load_profile("pmc", cfg, want)
    if gotp:
        kname, e = kernel_entry(gotp[1].get("kernels", {}), f"k_{names[dom]}")
        if e and e.get("SQ_INSTS_VALU"):
            valu, trans = e["SQ_INSTS_VALU"] * scale, e.get("SQ_INSTS_VALU_TRANS_F32", 0.0) * scale
            gui, us = e.get("GRBM_GUI_ACTIVE"), e.get("avg_us_in_pmc_pass")
            clock_ghz = gui / 8.0 / us / 1e3 if gui and us else NOMINAL_CLOCK_GHZ    # GRBM_GUI_ACTIVE sums the 8 XCDs (MI355X_MICROARCH.md, DVFS section)
            peak = N_SIMD * clock_ghz / VALU_CYCLES_PER_WAVE_INSTR                  # G wave-instructions / s
            ach = (valu + trans) / (avg_ms * 1e-3) / 1e9                            # a transcendental occupies the issue port twice as long
            wc = e.get("SQ_WAVE_CYCLES")
            valu_block = {
                "kernel": kname, "bound": "valu-issue", "achieved": round(ach, 2), "peak": round(peak, 2), "unit": "G wave-instr/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "measured_hbm_GBps": round(measured_gbps, 2) if measured_gbps else None,
                "measured_hbm_frac": round(measured_gbps / HBM_PEAK_GBS, 5) if measured_gbps else None,
                "avg_launch_ms": round(avg_ms, 5), "timed_launches": acc["kernel_timed"][dom],
                "launches_per_frame": round(acc["kernel_launches"][dom] / max(acc["samples"] / (float(W) * H * spp), 1e-9), 2),
                "valu_wave_instr_per_launch": valu, "transcendental_per_launch": trans,
                "clock_GHz_in_pmc_pass": round(clock_ghz, 3), "source": os.path.relpath(gotp[0], ROOT),
                # the instruction counts are the committed pass's: `profile_current` says whether that pass was taken on the library
                # that is being timed now (lr_build_info's content hash of csrc/* + flags, tools/build_id.py)
                "profile_build": gotp[1].get("workload", {}).get("build"), "library_build": LIB_BUILD[0],
                "profile_current": gotp[1].get("workload", {}).get("build") == LIB_BUILD[0],
                "counts_scaled_from_spp": scaled_from,
                "lanes_per_valu_instr": round(e["SQ_THREAD_CYCLES_VALU"] * scale / valu, 1) if e.get("SQ_THREAD_CYCLES_VALU") else None,
                # `frac` is issue-slot UTILISATION: a wave instruction counts the same with 64 or with 20 active lanes.  This is the
                # fraction of the chip's VALU lane-slots that did work: frac x active lanes per instruction / 64
                "lane_weighted_frac": round(ach / peak * (e["SQ_THREAD_CYCLES_VALU"] * scale / valu) / 64.0, 4) if e.get("SQ_THREAD_CYCLES_VALU") else None,
                "wave_cycles_parked": round(e["SQ_WAIT_ANY"] / wc, 3) if wc and e.get("SQ_WAIT_ANY") else None,
                "wave_cycles_issue_stalled": round(e["SQ_WAIT_INST_ANY"] / wc, 3) if wc and e.get("SQ_WAIT_INST_ANY") else None,
                "note": "the kernel keeps path state in registers / LDS and the scene in caches, so HBM is not its roof (measured_hbm_frac); "
                        "instructions per launch from the committed rocprofv3 pass of the same workload (deterministic per scene, film, spp "
                        "up to the seed), duration from this run's HIP events; peak = 1024 SIMDs x clock / 2 cycles per wave64 instruction",
            }

    # ---- SURVEY 8(d) algorithmic bytes (nominal for the one-launch kernels): 224 B per closest-hit segment, 108 B per shadow
    # ray, 16 B per child box tested (64-B 4-wide node), 48 B per primitive tested (1/64 when the rows come through the scalar
    # cache once per wave), 12 B per film pixel.  Box / primitive counts from one short counted render (LR_FLAG_COUNT).
    pc = desc.render_params(spp=min(32, spp), seed=77, integrator=integ, flags=abi.LR_FLAG_COUNT, path_slots=slots)
    scene.render(pc, tiles, n_tiles, out=canvas)
    sc = scene.stats()
    flat = desc.desc.n_prims <= 32
    queries = max(sc.segments + sc.shadow_rays, 1)
    v_per_q = (sc.node_visits + sc.shadow_node_visits) / queries
    t_per_q = (sc.prim_tests + sc.shadow_prim_tests) / queries
    scene_bytes_per_q = NODE_BYTES_PER_CHILD_BOX * v_per_q + PRIM_BYTES * t_per_q / (64.0 if flat else 1.0)
    bytes_per_sample = 224.0 * s_per + 108.0 * q_per + scene_bytes_per_q * (s_per + q_per) + 12.0 / spp
    if one_launch:
        # a frame whose chunk sums exceed 3 GiB is rendered in pixel bands, one launch each (lumilly_hip.hip): camera samples per
        # launch = the timed frames' samples over the timed frames' launches of this kernel
        units, unit_name = acc["samples"] / max(acc["kernel_launches"][dom], 1), "camera samples"
        bytes_per_launch = bytes_per_sample * units
    elif dom == abi.LR_K_SHADOW:
        units, unit_name = acc["shadow"] / max(acc["kernel_launches"][dom], 1), "shadow rays"
        per_unit = 48.0 + 12.0 + 4.0 + NODE_BYTES_PER_CHILD_BOX * sc.shadow_node_visits / max(sc.shadow_rays, 1) + PRIM_BYTES * sc.shadow_prim_tests / max(sc.shadow_rays, 1) / (64.0 if flat else 1.0)
        bytes_per_launch = per_unit * units
    elif dom == abi.LR_K_SHADE:
        units, unit_name = acc["segments"] / max(acc["kernel_launches"][dom], 1), "path vertices"
        bytes_per_launch = (96.0 + 80.0 + 4.0 + 64.0) * units
    else:
        units, unit_name = acc["segments"] / max(acc["kernel_launches"][abi.LR_K_TRACE], 1), "segments"
        per_unit = 32.0 + 8.0 + 4.0 + NODE_BYTES_PER_CHILD_BOX * sc.node_visits / max(sc.segments, 1) + PRIM_BYTES * sc.prim_tests / max(sc.segments, 1) / (64.0 if flat else 1.0)
        bytes_per_launch = per_unit * units
    nominal_gbps = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    cache_served = nominal_gbps > HBM_PEAK_GBS
    nominal = {
        "kernel": f"k_{names[dom]}", "bound": "hbm", "nominal": True, "achieved": round(nominal_gbps, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": None if cache_served else round(nominal_gbps / HBM_PEAK_GBS, 5), "cache_served": cache_served,
        "algorithmic_bytes_per_launch": round(bytes_per_launch, 0), "bytes_per_camera_sample": round(bytes_per_sample, 1),
        "units_per_launch": round(units, 1), "units": unit_name, "boxes_per_query": round(v_per_q, 2), "prim_tests_per_query": round(t_per_q, 2),
        "traffic": traffic, "measured_hbm_GBps": round(measured_gbps, 2) if measured_gbps else None,
        "note": "SURVEY 8(d): the bytes a wavefront design streams per unit (16 B per child box of the 64-B 4-wide node, 48 B per primitive). "
                + ("They exceed what HBM can deliver in the launch time: they are served by registers, LDS and caches, so no HBM frac is claimed. "
                   if cache_served else "")
                + "`traffic` is what really crosses the HBM interface per launch",
    }
    if valu_block:
        out["roofline"] = valu_block
        out["roofline_hbm_nominal"] = nominal
    elif not one_launch and measured_gbps:
        out["roofline"] = {
            "kernel": f"k_{names[dom]}", "bound": "hbm", "achieved": round(measured_gbps, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(measured_gbps / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_ms": round(avg_ms, 5), "timed_launches": acc["kernel_timed"][dom],
            "note": "streaming pipeline: MEASURED PMC bytes (FETCH_SIZE x2 + WRITE_SIZE) of this kernel per launch over this run's launch time",
        }
        out["roofline_hbm_nominal"] = nominal
    else:
        # no committed counter pass for this workload: only the nominal model can be printed, and it says so
        nominal["avg_launch_ms"] = round(avg_ms, 5)
        nominal["note"] = "NO committed PMC / traffic profile matches this workload -- nominal SURVEY 8(d) bytes only. " + nominal["note"]
        out["roofline"] = nominal
    out["kernels_ms_per_launch"] = {names[k]: round(acc["kernel_ms"][k] / acc["kernel_timed"][k], 5)
                                    for k in range(abi.LR_K_COUNT) if acc["kernel_timed"][k]}
    out["path_stats"] = {"segments_per_sample": round(s_per, 3), "shadow_rays_per_sample": round(q_per, 3), "pipeline": pipeline,
                         "path_slots": acc["path_slots"]}
    return out


def env_overrides():
    """Every LR_* variable set in this process's environment (name -> value)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("LR_")}


def setup_block(desc, scene, load_s, create_s):
    """What the reference lumps into `elapse` beside the render (main.rs:139-144) and prints as `bvh construction`
    (description.rs:67-73), timed separately: scene file + assets -> description, host SAH build, upload."""
    st = scene.stats()
    info = desc.bvh_info()
    return {
        "scene_load_s": round(load_s, 4),                        # host.Description(): TOML + OBJ/MTL/HDR parse, instancing, host SAH build
        "host_bvh_build_s": round(info["seconds"], 4),           # of which bvh.rs:69-127 (description.rs:67-73 prints this)
        "bvh_nodes": info["nodes"], "bvh_max_depth": info["max_depth"],
        "scene_create_s": round(create_s, 4),                    # lr_scene_create wall time (first call on a device includes context creation)
        "upload_ms": round(float(st.upload_ms), 3),              # of which host -> HBM copies + device-side table builds
        "device_bvh_build_ms": round(float(st.bvh_build_ms), 3),  # 0 = the host tree was uploaded (default)
        # the 4-wide tree on the device; nodes_without_distance_culling = what the sliver exemption (a heuristic) gives up
        "tree": scene.tree_info() if hasattr(scene, "tree_info") else None,
    }


def other_config_leg(abi, device, host, multigpu, cfg, steps, warmup, spp_override, args, flags):
    """One short single-GPU leg of another BASELINE config, after the headline's timed region: same lr_render path,
    HIP-event launch times, its own roofline block."""
    scene_file, W, H, stated_spp, integ, what, metric = CONFIGS[cfg]
    spp = spp_override or stated_spp
    t_l = time.perf_counter()
    desc = host.Description(os.path.join(ROOT, "scenes", scene_file))
    load_s = time.perf_counter() - t_l
    desc.set_resolution(W, H)
    integ_eff = desc.renderer.integrator if integ is None else integ
    t_c = time.perf_counter()
    scene = device.Scene(desc, device=0)
    create_s = time.perf_counter() - t_c
    setup = setup_block(desc, scene, load_s, create_s)
    tiles, n_tiles = multigpu.shard_tiles(W, H, args.tile, 0, 1)
    canvas = np.zeros((H, W, 3), dtype=np.float32)
    import torch
    for i in range(warmup):
        scene.render(desc.render_params(spp=spp, seed=1000 + i, integrator=integ, flags=flags), tiles, n_tiles, out=canvas)   # the same frame: buffers sized by spp (chunk sums) are allocated here, not in the timed region
    acc = new_acc(abi)
    torch.cuda.synchronize(0)
    t0 = time.perf_counter()
    for i in range(steps):
        scene.render(desc.render_params(spp=spp, seed=i, integrator=integ, flags=flags), tiles, n_tiles, out=canvas)
        add_stats(acc, scene.stats(), abi)
    torch.cuda.synchronize(0)
    elapsed = time.perf_counter() - t0
    assert acc["samples"] == W * H * spp * steps, f"{cfg}: device finished {acc['samples']} samples, expected {W * H * spp * steps}"
    # The Phong / Blinn-Phong rows are not finite everywhere IN THE REFERENCE: a sampled direction whose pdf underflows to 0 (phong.rs:47-68,
    # blinn_phong.rs:49-72 at alpha = 20) makes scene.rs:78-102 return 0 * c / 0 = NaN for the sample, and main.rs:92-121 folds it into the
    # pixel.  The oracle's film has the same pixels (the stated-size fixtures hold one each); counted here, not hidden.
    nonfinite = int((~np.isfinite(canvas)).any(axis=2).sum())
    assert nonfinite == 0 or cfg in ("c3p", "c3b"), f"{cfg}: non-finite film"
    integ_name = "pt-direct" if integ_eff == abi.LR_INTEGRATOR_PT_DIRECT else "pt"
    leg = {
        "metric": metric, "value": round(float(W) * H * spp * steps / elapsed / 1e6, 2), "unit": "Msamples/s", "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3),
        "config": {"workload": f"{cfg}: {scene_file} ({what}) {W}x{H} {spp} spp {integ_name}"
                               + ("" if spp == stated_spp else f" [{spp} of the stated {stated_spp} spp: samples are i.i.d., the rate does not depend on spp]"),
                   "baseline_config": cfg, "width": W, "height": H, "spp": spp, "stated_spp": stated_spp, "integrator": integ_name},
    }
    if nonfinite:
        leg["nonfinite_pixels"] = nonfinite       # (of W x H; the reference's own arithmetic, see above)
    leg["setup"] = setup
    leg.update(roofline_blocks(abi, cfg, scene_file, desc, scene, acc, W, H, spp, integ, tiles, n_tiles, canvas, 0, flags))
    scene.close()
    if not args.no_cpu_baseline:
        leg["cpu_baseline"] = cpu_baseline(desc, W, H, integ, args.leg_cpu_seconds)
    return leg


def c1_leg(abi, device, host, args):
    """BASELINE configs[0] at its stated size (256 x 256, 16 spp, pt: 1.05 M samples): the reference's own CPU-runnable case, so its
    line is the CPU's -- the oracle in both modes on the WHOLE job -- and the GPU renders the same frame only to be CHECKED against it
    (per-pixel bar, equal counters); no GPU `value` is claimed for a 1-M-sample job."""
    from oracle import binding as oracle
    scene_file, W, H, spp, integ, what, metric = CONFIGS["c1"]
    desc = host.Description(os.path.join(ROOT, "scenes", scene_file))
    desc.set_resolution(W, H)
    p = desc.render_params(spp=spp, seed=0, integrator=integ)
    cores = usable_cores()
    ref, st = oracle.render(desc, p, threads=cores, mode=oracle.BVH, pad=0.0, with_stats=True, fast=True)
    _, st2 = oracle.render(desc, p, threads=cores, mode=oracle.OWNBOX_ORDERED, pad=0.05, with_stats=True, fast=True)
    scene = device.Scene(desc, device=0)
    img = scene.render(p)
    gs = scene.stats()
    scene.close()
    err = float(np.nanmax(np.abs(img - ref) / np.maximum(1.0, np.abs(ref))))
    return {
        "metric": metric, "value": None, "unit": "Msamples/s",
        "config": {"workload": f"c1: {scene_file} ({what})", "baseline_config": "c1", "width": W, "height": H, "spp": spp, "integrator": "pt"},
        "cpu_baseline": {"value": round(st.samples / st.seconds / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
                         "sample": f"the whole job: {W}x{H} at {spp} spp ({st.samples} samples, {st.seconds:.2f} s), oracle in reference-literal BVH mode",
                         "optimized": {"value": round(st2.samples / st2.seconds / 1e6, 3), "unit": "Msamples/s",
                                       "sample": f"the whole job, ordered early-out traversal with the leaf's own-box test ({st2.seconds:.2f} s)"}},
        "gpu_check": {"linf_rel_vs_oracle": err, "within_1e-4": bool(err < 1e-4), "device_ms": round(float(gs.render_ms), 3),
                      "counters_equal": (int(gs.samples), int(gs.segments), int(gs.shadow_rays)) == (int(st.samples), int(st.segments), int(st.shadow_rays))},
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    overrides = env_overrides()
    blocking = [k for k in overrides if k in PRODUCT_ENV]
    if blocking and not args.allow_overrides:
        raise SystemExit(f"bench.py: {', '.join(blocking)} set in the environment: these change what the library runs, so the number would not be "
                         "the product's. Unset them, or pass --allow-overrides (they are then recorded in the line's env_overrides).")

    import torch
    from lumillyrender_amd import abi, device, host, multigpu
    if args.tile <= 0:
        args.tile = host.default_tile()
    LIB_BUILD[0] = device.build_id()

    scene_file, W, H, base_spp, integ, what, metric = CONFIGS[args.config]
    scene_file = args.scene or scene_file
    W, H = args.width or W, args.height or H
    base_spp = args.spp or base_spp
    stated = (scene_file, W, H, base_spp) == CONFIGS[args.config][:4]

    dist = None
    host_group = None
    dev_group = None
    barrier_kind = "device synchronize (single process)"
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"     # the env switch exercises the RCCL path on a 1-GPU box
    if use_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.same_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        # host side (film assembly, max of the timings): gloo, the default group.  Device side (the barrier that
        # brackets the timed region): an RCCL group when it comes up; a machine where it does not still gets its
        # numbers through the gloo barrier + device synchronisation.
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        barrier_kind = "gloo barrier + device synchronize"
        if "nccl" in args.backend:
            try:
                import datetime
                dev_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
                probe = torch.ones(1, device=f"cuda:{local_rank}")
                dist.all_reduce(probe, group=dev_group)
                torch.cuda.synchronize(local_rank)
                assert int(probe.item()) == world
                barrier_kind = "RCCL all-reduce + device synchronize"
            except Exception as e:                                     # noqa: BLE001 -- report and fall back
                print(f"[bench] rank {rank}: RCCL group unavailable ({type(e).__name__}: {e}); using the gloo barrier", file=sys.stderr, flush=True)
                dev_group = None
    dev_index = local_rank if world > 1 else 0

    spp = base_spp * (world if args.scaling == "weak" else 1)
    t_l = time.perf_counter()
    desc = host.Description(os.path.join(ROOT, "scenes", scene_file))
    load_s = time.perf_counter() - t_l
    desc.set_resolution(W, H)
    integ_eff = desc.renderer.integrator if integ is None else integ
    t_c = time.perf_counter()
    scene = device.Scene(desc, device=dev_index)            # scene resident in HBM from here on
    create_s = time.perf_counter() - t_c
    setup = setup_block(desc, scene, load_s, create_s)
    tiles, n_tiles = multigpu.shard_tiles(W, H, args.tile, rank, world)
    flags = (0 if args.no_profile else abi.LR_FLAG_PROFILE) | (abi.LR_FLAG_STREAMING if args.streaming else 0) | (abi.LR_FLAG_RESIDENT if args.resident else 0)
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

    walls = {"call": 0.0, "wait": 0.0}                        # this rank's wall seconds inside lr_render / inside the per-frame barrier (timed steps only)

    def step(i, timed=False):
        params = desc.render_params(spp=spp, seed=i, integrator=integ, flags=flags, path_slots=args.slots)
        t_a = time.perf_counter()
        scene.render(params, tiles, n_tiles, out=canvas)      # blocks until this rank's tiles are in the (shared) host film
        t_b = time.perf_counter()
        st = scene.stats()
        shared_film.collect()                                   # every rank's tiles are in the shared film; a barrier publishes it to rank 0
        if timed:
            walls["call"] += t_b - t_a; walls["wait"] += time.perf_counter() - t_b
        return st

    for i in range(args.warmup):
        step(1000 + i)
    acc = new_acc(abi)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        add_stats(acc, step(i, True), abi)
    barrier()
    elapsed = time.perf_counter() - t0
    my_pixels = sum(tiles[i].w * tiles[i].h for i in range(n_tiles))
    assert acc["samples"] == my_pixels * spp * args.steps, f"device finished {acc['samples']} samples, expected {my_pixels * spp * args.steps}"
    rank_ms = [acc["render_ms"] / max(args.steps, 1)]
    rank_setup = [setup]
    n_st = max(args.steps, 1)
    dom_k = max(range(abi.LR_K_COUNT), key=lambda k: acc["kernel_ms"][k] / max(acc["kernel_timed"][k], 1) * acc["kernel_launches"][k])
    # what one rank's frame consists of (main.rs:129-132: the reference drains a channel; here every rank renders, reads back and meets
    # the others): per-step means of this rank
    mine = {"render_ms": acc["render_ms"] / n_st,                                  # device work of lr_render (HIP events)
            "dominant_kernel": abi.LR_KERNEL_NAMES[dom_k],
            "dominant_kernel_ms": acc["kernel_ms"][dom_k] / max(acc["kernel_timed"][dom_k], 1) * (acc["kernel_launches"][dom_k] / n_st),
            "render_call_ms": walls["call"] / n_st * 1e3,                          # wall time inside lr_render: launch + device work + read-back + scatter into the film
            "readback_ms": max(walls["call"] / n_st * 1e3 - acc["render_ms"] / n_st, 0.0),
            "barrier_wait_ms": walls["wait"] / n_st * 1e3,                         # waiting for the slowest rank of the frame
            "host_bvh_build_s": setup["host_bvh_build_s"], "upload_ms": setup["upload_ms"],
            "pixels": my_pixels, "rays": (acc["segments"] + acc["shadow"]) // n_st}
    rank_rows = [mine]
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        all_ms = [None] * world
        dist.all_gather_object(all_ms, rank_ms[0])
        rank_ms = all_ms
        all_setup = [None] * world
        dist.all_gather_object(all_setup, setup)
        rank_setup = all_setup
        all_rows = [None] * world
        dist.all_gather_object(all_rows, mine)
        rank_rows = all_rows

    total_samples = float(W) * H * spp * args.steps
    value = total_samples / elapsed / 1e6

    out = None
    if rank == 0:
        # (the Phong / Blinn-Phong rows hold the reference's own NaN pixels: see other_leg)
        nonfinite = int((~np.isfinite(np.asarray(canvas))).any(axis=2).sum())
        assert nonfinite == 0 or args.config in ("c3p", "c3b"), "non-finite film"
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
                "parallelism": f"pixel tiles dealt diagonally ((i + k j) mod world) over {world} GPU(s), replicated scene, film assembled in host shared memory, no collective on the data path",
                "path_slots": acc["path_slots"], "pipeline": PIPELINE_NAMES.get(acc["pipeline"], "?"),
            },
            "rank_render_ms": {"max": round(max(rank_ms), 3), "min": round(min(rank_ms), 3)},
            # every rank's share of a frame, one list entry per rank (per-step means over the timed steps): a SCALE run that falls short
            # says where -- an uneven deal (rays), a slow device (render_ms at equal rays), the read-backs (readback_ms), or waiting
            "ranks": {k: [round(r[k], 3) if isinstance(r[k], float) else r[k] for r in rank_rows] for k in mine},
            "barrier": barrier_kind,
            # setup costs, NOT part of `value` (the scene is resident before the timed region): rank 0's in full, every rank's upload beside it
            "setup": setup,
            "rank_upload_ms": [round(x["upload_ms"], 3) for x in rank_setup],
            "rank_host_bvh_build_s": [x["host_bvh_build_s"] for x in rank_setup],
            "env_overrides": overrides,
        }
        if nonfinite:
            out["nonfinite_pixels"] = nonfinite
        if world == 1:
            out.update(roofline_blocks(abi, args.config, scene_file, desc, scene, acc, W, H, spp, integ, tiles, n_tiles, canvas, args.slots, flags))
    scene.close()
    shared_film.close()
    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, W, H, integ, args.cpu_seconds)
        default_run = stated and args.config == "c2" and not (args.streaming or args.resident)
        if default_run and not args.no_other_configs:
            # the other BASELINE configs, AFTER the headline's timed region (they do not touch `value`)
            legs = {}
            try:
                legs["c1"] = c1_leg(abi, device, host, args)
            except Exception as e:                                      # noqa: BLE001
                legs["c1"] = {"error": f"{type(e).__name__}: {e}"}
            for cfg, steps, warmup, spp_o in OTHER_LEGS:
                try:
                    legs[cfg] = other_config_leg(abi, device, host, multigpu, cfg, steps, warmup, spp_o, args, abi.LR_FLAG_PROFILE)
                except Exception as e:                                  # noqa: BLE001 -- e.g. generated assets missing: say so, keep the headline
                    legs[cfg] = {"error": f"{type(e).__name__}: {e}"}
            out["other_configs"] = legs
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
