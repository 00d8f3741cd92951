#!/usr/bin/env python3
"""Benchmark of the GPU path tracer on the headline configuration.

One step = one full render of the workload (every sample of every pixel) through rt_render, the per-rank sum buffers
reduced to rank 0 with torch.distributed (RCCL) when more than one GPU takes part, and the 8-bit image resolved on
rank 0.  The frame is split by sample range (rank r renders samples [r*S/N, (r+1)*S/N) of every pixel): total work is
fixed, so scaling is "strong".  value = rays of all ranks (closest-hit + shadow queries) per second of the slowest rank.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, width, height, spp, shader)
    "c1_cornell": ("cornell", 512, 512, 16, "mis"),
    "c2_whitted": ("whitted", 1600, 1600, 16, "whitted"),
    "c3_utah": ("utah", 1920, 1080, 256, "mis"),
    "c4_miku": ("miku", 1920, 1080, 1024, "mis"),
    "c5_composite": ("composite", 3840, 2160, 4096, "mis"),
    "head": ("head", 800, 800, 800, "mis"),
}


def launch_ranks(n: int) -> int:
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment), same arguments."""
    import socket
    import subprocess

    import torch  # (device_count() starts the HIP runtime in THIS process; harmless: the ranks are fresh child processes and this one never renders)

    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def source_id() -> str:
    """Hash of the sources the device library is built from: ties a cached counter file to the build it was taken on."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer-rs_amd", "csrc")
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3_utah", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override the workload's samples per pixel (invalidates the headline number)")
    ap.add_argument("--shard", default="spp", choices=["spp", "tile"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-count-pass", action="store_true", help="skip the untimed BVH work-counting pass (for PMC profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the baseline sample")
    ap.add_argument("--roulette", action="store_true", help="RT_FLAG_RUSSIAN_ROULETTE, an option the reference does not have: NOT the headline configuration")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank: rehearses the N>1 code path")
    args = ap.parse_args()

    # --gpus N without a launcher around this script: start the N ranks here, before this process touches torch or HIP
    # (a process that has initialised the GPU must not be replaced or forked), and relay rank 0's JSON line.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}", file=sys.stderr)
        return 2

    # The renderer overlaps two streams, torch and RCCL bring theirs; HIP's default of 4 hardware queues per process
    # makes streams share queues (and lose the overlap) once RCCL is up.  Must be set before the HIP runtime starts.
    try:
        if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
    except ValueError:
        pass

    import numpy as np
    import torch
    import torch.distributed as dist

    rt = importlib.import_module("raytracer-rs_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        # RCCL creates its streams at the first collective.  Do that before the renderer creates its two streams: a
        # process has few hardware queues (GPU_MAX_HW_QUEUES), and streams that share one do not overlap.
        warm = torch.zeros(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(warm)
        torch.cuda.synchronize()
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    scene_name, W, H, spp, shader_name = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    shader = {"mis": rt.SHADER_MIS, "whitted": rt.SHADER_WHITTED, "path": rt.SHADER_PATH, "normal": rt.SHADER_NORMAL}[shader_name]

    hs = rt.HostScene(scene_name, W / H)
    t_cold = time.perf_counter()
    ctx = rt.Context(local_rank)
    t0 = time.perf_counter()
    ds = rt.DeviceScene(ctx, hs.desc)
    scene_ms = (time.perf_counter() - t0) * 1e3

    plan = rt.plan_shards(args.shard, rank, world, spp, tile_size=64)
    extra_flags = rt.FLAG_RUSSIAN_ROULETTE if args.roulette else 0
    params = plan.apply(rt.default_params(W, H, spp, shader=shader, seed=1, flags=rt.FLAG_TIMING | extra_flags))

    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    hip = rt.hip_lib()

    def step():
        # Three streams take part: torch's current stream (zeroing, RCCL), and the context's stream (rt_render and
        # rt_resolve_device, both of which return with their work complete).  Each hand-over is a full synchronisation
        # of the producing stream: render -> reduce -> resolve, one after the other.
        accum.zero_()
        torch.cuda.current_stream().synchronize()
        st = ds.render_device(hs.camera, params, accum.data_ptr())
        if use_dist:
            rt.reduce_accum(accum, dst=0)
            torch.cuda.current_stream().synchronize()  # the reduce runs on torch's stream, the resolve on the context's
        if rank == 0:
            rc = hip.rt_resolve_device(ctx._h, accum.data_ptr(), W * H, spp, rgb8.data_ptr())
            assert rc == 0, hip.rt_last_error()
        return st

    # cold start: context, scene build + upload, and the first render with its queue allocation, on this rank
    st_cold = step()
    cold_ms = (time.perf_counter() - t_cold) * 1e3

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    fence()
    elapsed = time.perf_counter() - t0

    # aggregate over ranks: slowest rank's time, everyone's rays
    rays_local = float(sum(s.rays_closest + s.rays_shadow for s in stats))
    cam_local = float(sum(s.camera_samples for s in stats))
    closest_local = float(sum(s.rays_closest for s in stats))
    agg = torch.tensor([rays_local, cam_local, closest_local], dtype=torch.float64, device=dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_rays, total_cam, total_closest = agg.tolist()
    elapsed = float(tmax.item())

    result = None
    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = elapsed / steps * 1e3
        mrays = total_rays / elapsed / 1e6

        # ---- roofline of the dominant kernel, from this rank's timed steps (HIP events on the render stream)
        # Algorithmic HBM bytes per unit: what one launch must stream to and from HBM.  BVH nodes and
        # triangles are NOT in these figures: the whole scene (a few MB) stays in L2 / Infinity Cache / LDS after the first
        # touch; their per-ray volume is reported beside it as cache_bytes_per_ray.
        closest = sum(s.rays_closest for s in stats)
        shadow = sum(s.rays_shadow for s in stats)
        cams = sum(s.camera_samples for s in stats)
        children_per_node = (closest - cams) / max(1, closest)
        inline = sum(s.rays_shadow_inline for s in stats)  # occlusion queries k_shade answered itself: no queue traffic
        queued = shadow - inline
        shadow_per_node = queued / max(1, closest)
        inline_per_node = inline / max(1, closest)
        # nodes that carried their path's light: 16 B more read, 16 B more written for the successor; what such paths gather
        # (the in-place queries' contributions among it) reaches the pixel as ONE add per path instead of one per query
        carrying = sum(s.nodes_carrying for s in stats)
        carry_per_node = carrying / max(1, closest)
        adds_per_node = (cams / max(1, closest)) if carrying else inline_per_node
        stage_ms = {k: sum(getattr(s, "ms_" + k) for s in stats) / steps for k in ("extend", "shade", "shadow", "raygen")}
        # mean BVH work per ray, counted in a separate (untimed, slower) pass over 1 sample per pixel
        nodes_per_ray = tris_per_ray = 0.0
        walk_per_ray = {}  # kernel -> (tree nodes visited, tree-leaf triangles tested) per ray of that kernel
        if not args.no_count_pass:
            cp = rt.default_params(W, H, spp, shader=shader, seed=1, flags=rt.FLAG_COUNT_WORK | extra_flags)
            cp.spp_begin, cp.spp_end = 0, 1
            scratch = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            cst = ds.render_device(hs.camera, cp, scratch.data_ptr())
            n_rays_counted = max(1, cst.rays_closest + cst.rays_shadow)
            nodes_per_ray = cst.nodes_visited / n_rays_counted
            tris_per_ray = cst.tris_tested / n_rays_counted
            c_queued = max(1, cst.rays_shadow - cst.rays_shadow_inline)
            walk_per_ray = {"k_extend": ((cst.nodes_visited - cst.nodes_visited_shadow) / max(1, cst.rays_closest),
                                         (cst.tris_tested - cst.tris_tested_shadow - cst.tris_scanned) / max(1, cst.rays_closest)),
                            "k_shadow": (cst.nodes_visited_shadow / c_queued, cst.tris_tested_shadow / c_queued)}
        d = hs.desc
        cache_bytes_per_ray = 32 * nodes_per_ray + 48 * tris_per_ray + 16 * d.n_sph + 20 * d.n_cyl
        per_unit = {  # kernel: (units over the timed steps, algorithmic HBM bytes per unit, launches, total ms)
            "k_extend": (closest, 32 + 8, sum(s.launches_extend for s in stats), sum(s.ms_extend for s in stats)),
            "k_shade": (closest, 48 + 8 + 48 * children_per_node + 40 * shadow_per_node + 12 * adds_per_node + 32 * carry_per_node, sum(s.launches_shade for s in stats),
                        sum(s.ms_shade for s in stats)),
            "k_shadow": (queued, 40 + 12, sum(s.launches_shadow for s in stats), sum(s.ms_shadow for s in stats)),
            # camera rays written (48 B each).  Its launches run beside the previous iteration's k_shadow on the second stream:
            # alone (RT_OVERLAP=0) the kernel writes at 6.2 TB/s, see profiles/r3_raygen_alone_vs_overlapped.txt
            "k_raygen": (cams, 48, sum(s.launches_raygen for s in stats), sum(s.ms_raygen for s in stats)),
        }
        # HBM traffic from the PMC counters is not collected by this run: it comes from the committed counter passes of
        # this same command, and only when they were taken on a library built from the sources in this tree
        traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        measured_traffic = {}
        traffic_note = "no counter file"
        if os.path.exists(traffic_file):
            with open(traffic_file) as f:
                tj = json.load(f)
            if tj.get("workload") != args.workload or args.spp:
                traffic_note = "counter file is for another workload"
            elif tj.get("source_id") != source_id():
                traffic_note = "counter file was taken on another build"
            else:
                measured_traffic = tj.get("bytes_per_launch", {})
                traffic_note = "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command on this build)"
        kernels = {}
        for name, (units, bpu, launches, ms) in per_unit.items():
            if launches and ms > 0:
                avg_ms = ms / launches
                achieved = bpu * (units / launches) / (avg_ms * 1e-3) / 1e9
                kernels[name] = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(achieved / 8000.0, 4), "traffic": measured_traffic.get(name), "avg_launch_ms": round(avg_ms, 4),
                                 "launches_per_step": launches / steps, "units_per_launch": round(units / launches, 1),
                                 "bytes_per_unit": round(bpu, 1), "ms_per_step": round(ms / steps, 2)}
        # ---- the walk kernels' own roofline: per-lane ("divergent") 16-byte loads per second.  A visited tree node is one such
        # load, a tested leaf triangle three (its 48-byte record); the scene (a few MB) stays in L2, so these never become
        # HBM traffic and the HBM figure above says nothing about a walk.  A CU's vector cache keeps nothing of it (measured hit
        # rate 1 %: thousands of rays per CU, each after another cache line), so every fetch is a line request to the L2, and
        # loads of one lane that fall into one line (a node and its successor, the three parts of a triangle record) share a
        # request.  Peak = what tools/ta_rate measures on this GPU for a table of the scene's footprint with exactly those
        # access shapes: the best of a dependent chain of adjacent load pairs, independent 32-byte records (two loads) and
        # independent packed 48-byte records (three loads), every lane at a random record of its own.
        info = ds.info
        footprint = info.n_nodes * info.n_orders * 16 + info.n_triangles * 48
        ta = None
        ta_path = os.path.join(ROOT, "tools", "ta_rate")
        if walk_per_ray and info.n_nodes > 0 and os.path.exists(ta_path):
            import subprocess
            try:
                out = subprocess.run([ta_path, "--json", "--footprint", str(max(footprint, 1 << 16))], capture_output=True, text=True, timeout=120)
                ta = json.loads(out.stdout)
            except Exception as e:  # the figure is then simply missing
                print(f"bench.py: tools/ta_rate failed: {e}", file=sys.stderr)
        if ta:
            shapes = ("divergent 16 B, dependent chain of load PAIRS", "records of 32 B, one lane each", "records of 48 B packed, one lane each")
            rows16 = [r for r in ta["rows"] if r["bytes"] == 16 and r["what"].startswith(shapes) and r["active_lanes"] == 64]
            peak_row = max(rows16, key=lambda r: r["Glane_per_s"])
            for name, (n_node, n_tri) in walk_per_ray.items():
                if name not in kernels or n_node <= 0:
                    continue
                k = kernels[name]
                acc_per_ray = n_node + 3.0 * n_tri
                achieved = acc_per_ray * k["units_per_launch"] / (k["avg_launch_ms"] * 1e-3) / 1e9
                k["l1_access"] = {"bound": "l1_access", "kernel": name, "achieved": round(achieved, 1), "peak": round(peak_row["Glane_per_s"], 1),
                                  "unit": "G per-lane 16-byte loads/s", "frac": round(achieved / peak_row["Glane_per_s"], 4), "traffic": None,
                                  "accesses_per_ray": round(acc_per_ray, 2), "nodes_per_ray": round(n_node, 2), "leaf_triangles_per_ray": round(n_tri, 2),
                                  "peak_from": f"tools/ta_rate, table of {ta['footprint_bytes']} bytes: {peak_row['what']}, {peak_row['waves_per_simd']} waves per SIMD",
                                  "avg_launch_ms": k["avg_launch_ms"], "units_per_launch": k["units_per_launch"]}
        roofline = None
        if kernels:
            dom = max(kernels.values(), key=lambda k: k["ms_per_step"])
            if "l1_access" in dom and dom["l1_access"]["frac"] > dom["frac"]:
                # a walk kernel dominates: its bound is the rate of per-lane loads, not HBM
                hbm = {kk: dom[kk] for kk in ("achieved", "peak", "unit", "frac", "bytes_per_unit")}
                dom = dict(dom, **dom["l1_access"])
                dom["hbm"] = hbm
                dom.pop("l1_access", None)
            roofline = dict(dom)
            roofline["note"] = ("HBM roofline of the kernel with the largest share of the step; traversal kernels are bound by divergence and "
                                "L1/L2 latency, not HBM: BVH/triangle fetches (cache_bytes_per_ray) never leave the caches.  k_shadow of one "
                                "iteration runs on a second stream beside k_extend/k_shade of the next, so the kernels' durations overlap "
                                "(their sum exceeds the step) and each includes the slowdown from sharing the GPU")
            total_bytes = sum(units * bpu for units, bpu, _, _ in per_unit.values()) / steps
            roofline["whole_step"] = {"algorithmic_bytes": round(total_bytes), "achieved": round(total_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                      "frac": round(total_bytes / (ms_per_step * 1e-3) / 1e9 / 8000.0, 4), "unit": "GB/s"}
            # The per-ray figure that also counts geometry fetches (32 B ray + 8 B hit + 32 B per visited node + 36 B per
            # triangle test + 16 / 24 B per sphere / cylinder, 64 B of path state per closest-hit ray, 12 B per camera sample)
            # and the matching arithmetic (25 flop per node, 51 per triangle, 30 / 60 per sphere / cylinder, 150 per shaded
            # hit): most of those bytes are served by LDS / L1 / L2, so their rate may exceed the HBM peak; the flop rate is
            # priced against the non-fused FP32 vector rate (157.3 / 2 TFLOP/s: the kernels are built without contraction).
            if nodes_per_ray > 0:
                geo_b = 32 * nodes_per_ray + 36 * tris_per_ray + 16 * d.n_sph + 24 * d.n_cyl
                geo_f = 25 * nodes_per_ray + 51 * tris_per_ray + 30 * d.n_sph + 60 * d.n_cyl
                frame_b = total_rays * (40 + geo_b) + total_closest * 64 + total_cam * 12  # all ranks, all timed steps
                frame_f = total_rays * geo_f + total_closest * 150
                sec = ms_per_step * 1e-3 * steps
                roofline["with_geometry_fetches"] = {
                    "bytes_per_ray": round(40 + geo_b, 1), "flops_per_ray": round(geo_f, 1),
                    "achieved_GBps": round(frame_b / sec / 1e9, 1), "achieved_TFLOPs": round(frame_f / sec / 1e12, 2),
                    "valu_peak_TFLOPs": 78.65, "valu_frac": round(frame_f / sec / 1e12 / 78.65, 4)}
            roofline["traffic_source"] = traffic_note
            roofline["cache_bytes_per_ray"] = round(cache_bytes_per_ray, 1)
            roofline["nodes_per_ray"] = round(nodes_per_ray, 2)
            roofline["tris_per_ray"] = round(tris_per_ray, 2)
            roofline["all_kernels"] = {k: {kk: v[kk] for kk in ("achieved", "frac", "avg_launch_ms", "bytes_per_unit", "ms_per_step", "traffic", "l1_access") if kk in v}
                                       for k, v in kernels.items()}
            if ta:
                both = sum((walk_per_ray[n][0] + 3.0 * walk_per_ray[n][1]) * kernels[n]["units_per_launch"] * kernels[n]["launches_per_step"]
                           for n in walk_per_ray if n in kernels)
                roofline["walk_kernels_together"] = {
                    "note": "k_extend and k_shadow run side by side on two streams, so each one's launch time includes the other's share of the GPU: "
                            "per-lane loads of both over the whole step",
                    "achieved": round(both / (ms_per_step * 1e-3) / 1e9, 1), "peak": round(peak_row["Glane_per_s"], 1),
                    "frac": round(both / (ms_per_step * 1e-3) / 1e9 / peak_row["Glane_per_s"], 4), "unit": "G per-lane 16-byte loads/s"}
                roofline["ta_rate"] = ta

        # ---- CPU baseline: the oracle (a restatement of the reference, not the Rust binary) on a bounded sample
        cpu = None
        pruned = None
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py as orc

            osc = orc.OracleScene(hs.desc, 1)
            threads = min(orc.num_threads(), int(os.environ.get("OMP_NUM_THREADS", "16")))
            # calibrate on a coarse frame, then size the sample (same scene, camera and frame, first k samples per pixel
            # of a sub-sampled frame) for about cpu_seconds of work on all threads
            cw, ch = max(8, W // 8), max(8, H // 8)
            pp = rt.default_params(cw, ch, spp, shader=shader, seed=1)
            pp.spp_begin, pp.spp_end = 0, 1
            t0 = time.perf_counter()
            _, cs = osc.render(hs.camera, pp, threads=threads)
            dt = max(time.perf_counter() - t0, 1e-3)
            scale = max(1.0, 0.6 * args.cpu_seconds / dt)
            # grow the frame first (up to the full frame), then the number of samples
            f = min(8.0, scale ** 0.5)
            sw, sh_ = min(W, int(cw * f)), min(H, int(ch * f))
            k = int(max(1, min(spp, round(scale / ((sw * sh_) / (cw * ch))))))
            pp = rt.default_params(sw, sh_, spp, shader=shader, seed=1)
            pp.spp_begin, pp.spp_end = 0, k
            t0 = time.perf_counter()
            _, cs = osc.render(hs.camera, pp, threads=threads)
            dt = time.perf_counter() - t0
            cpu_rays = cs.rays_closest + cs.rays_shadow
            # the reference itself is single-threaded (src/bin/main.rs): the same frame, fewer samples, on one thread
            k1 = max(1, int(round(k * 0.4 * args.cpu_seconds / max(dt, 1e-3) / max(1, threads))))
            p1 = rt.default_params(sw, sh_, spp, shader=shader, seed=1)
            p1.spp_begin, p1.spp_end = 0, min(k, k1)
            t0 = time.perf_counter()
            _, cs1 = osc.render(hs.camera, p1, threads=1)
            dt1 = time.perf_counter() - t0
            cpu = {"value": round(cpu_rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": f"{scene_name} {sw}x{sh_}, samples 0..{k} of {spp} per pixel, {cpu_rays} rays in {dt:.1f} s "
                             f"(C++ restatement of the reference with its median-split BVH; counts every ray the reference traces)",
                   "single_thread": {"value": round((cs1.rays_closest + cs1.rays_shadow) / dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                                     "sample": f"samples 0..{p1.spp_end} of the same frame, {cs1.rays_closest + cs1.rays_shadow} rays in {dt1:.1f} s"}}
            # The GPU does not trace rays whose weight is exactly zero (the reference traces them and multiplies by zero):
            # the same sample on the GPU gives the fraction of the reference's rays that `value` does not count.
            scratch = torch.zeros((sh_, sw, 3), dtype=torch.float32, device=dev)
            torch.cuda.current_stream().synchronize()
            gst = ds.render_device(hs.camera, pp, scratch.data_ptr())
            gpu_rays = gst.rays_closest + gst.rays_shadow
            pruned = {"fraction": round(1.0 - gpu_rays / max(1, cpu_rays), 5), "gpu_rays": gpu_rays, "reference_rays": cpu_rays,
                      "note": "rays of zero weight the GPU skips; value counts only rays actually traced"}

        result = {
            "metric": "Mrays/sec (whole node) + time-to-image, 1080p 256spp, 1/2/4/8 MI355X",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {scene_name} {W}x{H} {spp} spp, {shader_name} shader" + (" + Russian roulette (not the reference's estimator)" if args.roulette else ""),
                       "shard": args.shard,
                       "rays_per_step": total_rays / steps, "camera_samples_per_step": total_cam / steps,
                       "time_to_image_ms": round(ms_per_step, 2), "scene_upload_ms": round(scene_ms, 2),
                       "cold_time_to_image_ms": round(cold_ms, 1), "device_bytes": int(stats[-1].device_bytes) if stats else 0,
                       "zero_weight_rays_pruned": pruned,
                       "stage_ms_per_step_rank0": {k: round(v, 2) for k, v in stage_ms.items()}},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(result), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
