#!/usr/bin/env python3
"""Rasteriser fwd+bwd throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step is one forward+backward render of BASELINE config 5 (2M Gaussians, 1920x1080, synthetic):
preprocess, binning, blend, blend backward, per-Gaussian backward with the pose Jacobian.  With
N > 1 every rank (one per GPU) renders its own keyframe of the mapping window against the same
Gaussians and the ranks all-reduce the Gaussian gradients over RCCL (weak scaling: one 1080p keyframe
per GPU).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

``python bench.py --gpus N`` works as typed: when it is not already running under torch.distributed.run
it starts N fresh rank processes itself (before anything touches the GPU) and exits with their code; the
driver's ``python -m torch.distributed.run ... bench.py --gpus N`` form is used as it is.

``--workload c4`` times BASELINE config 4 instead: a mapping iteration over an 8-keyframe window at Replica
resolution, keyframes sharded over the ranks (`monogs_amd.mapping.WindowMapper`; strong scaling).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default c5: 6000 = ~6 s of GPU time at N=1, longer than the 5 s period of an outside "
                         "utilisation sampler; c4: 200 -- its steps are optimiser iterations, so a long run is a different "
                         "map at the end than at the start)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-steps", type=int, default=60,
                    help="c5: un-timed steps of the same workload run BEFORE the --warmup steps, so that a short run is timed "
                         "with the chip in the power state a long one reaches by itself: the two VALU-bound blend kernels get "
                         "~6 %% faster over the first ~40 ms of sustained load after process start (profiles/r04_warmup_probe.txt, "
                         "profiles/r04_bench_settle.log).  0 = off.  Reported in the line as config.settle_steps; the timed "
                         "region is still exactly --steps steps between two fences")
    ap.add_argument("--gaussians", type=int, default=2_000_000)
    ap.add_argument("--intrinsics", default="davis_1080p")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-slam", action="store_true", help="skip the short tracking+mapping run (N=1 only)")
    ap.add_argument("--exact-count", action="store_true",
                    help="read the instance count back every forward (upstream behaviour) instead of capacity mode")
    ap.add_argument("--profile-steps", type=int, default=9,
                    help="steps timed per stage with HIP events, outside the timed region (median reported; an event\n"
                         "between every two stages costs a few us of stream time each, so the stages add up to a few\n"
                         "percent MORE than ms_per_step)")
    ap.add_argument("--trace-region", action="store_true",
                    help="diagnostic (runs of <= 64 steps): an event record after every step of the TIMED region and the allocator's "
                         "reservation per step, logged to stderr; off by default so that the timed region holds nothing but the steps "
                         "and the <= 8 blend-kernel probes")
    ap.add_argument("--trace-steps", type=int, default=0,
                    help="diagnostic: after the timed region, N more steps behind a fence with a HIP event after each one; prints "
                         "the per-step device time (how long a short run takes to reach the steady state)")
    ap.add_argument("--workload", choices=("c5", "c4"), default="c5",
                    help="c5: one 1080p keyframe per GPU (headline); c4: 8-keyframe Replica mapping window sharded over the GPUs")
    ap.add_argument("--window", type=int, default=8, help="c4: keyframes in the mapping window")
    ap.add_argument("--eager", action="store_true", help="c4: no hipGraph replay (every iteration launched from Python)")
    ap.add_argument("--exchange", choices=("bucket", "per_keyframe"), default="bucket",
                    help="c4, N > 1: one all-reduce of the summed gradient bucket per iteration (default), or one per owned keyframe "
                         "issued behind its backward while the next keyframe renders (WindowMapper.exchange)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 6000 if args.workload == "c5" else 200
    return args


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start N fresh rank processes (this process has not
    touched the GPU and never will) and hand their exit code back.  Rank 0's JSON line goes straight to our stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def csrc_hash() -> str:
    """Identity of the kernel sources this run was built from (the .git directory does not travel to the GPU box):
    sha256 over monogs_amd/csrc/* and include/*.  profiles/*.json carry the same stamp; stale files are ignored."""
    import hashlib
    h = hashlib.sha256()
    for d in ("monogs_amd/csrc", "include"):
        for name in sorted(os.listdir(os.path.join(ROOT, d))):
            if name.endswith((".hip", ".h", "Makefile")):
                h.update(name.encode())
                h.update(open(os.path.join(ROOT, d, name), "rb").read())
    return h.hexdigest()[:16]


def stamped_profile(name: str):
    """profiles/<name> if it was collected on exactly these kernel sources, else None."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(path))
    except Exception:
        return None
    return d if d.get("csrc_sha256") == csrc_hash() else None


def cpu_baseline():
    """The oracle (float32 PyTorch-CPU autograd rasteriser) on a density-preserving crop of the workload: same focal
    length and generator, 1/4 of the pixels and of the Gaussians; one un-timed warm-up pass (thread pool, allocator),
    then the median of three timed fwd+bwd passes (~20-30 s of CPU work in all)."""
    import torch

    from monogs_amd.synthetic import make_scene, scene_settings
    from oracle import OracleSettings, rasterize_autograd

    intr = dict(fx=960.0, fy=960.0, cx=480.0, cy=270.0, W=960, H=540)
    sc = make_scene(500_000, intr, seed=2)
    st = scene_settings(sc, OracleSettings)
    inp = dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=sc.colors,
               scales=sc.scales.repeat(1, 3), rotations=sc.rotations)
    # the box's CPU share, not the host's core count (over-subscribing OpenMP stalls tiny ops)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        rasterize_autograd(inp, st, sc.grad_color, sc.grad_depth, dtype=torch.float32)
        if i:
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    return {"value": round(intr["W"] * intr["H"] / 1e6 / dt, 5), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": f"500k Gaussians, 960x540 crop of the 1080p workload (same focal length, same per-pixel "
                      f"density), fwd+bwd, float32, 1 warm-up + median of 3: {dt:.2f} s "
                      f"(min {min(times):.2f}, max {max(times):.2f})"}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def bench_c4(args, rank, world, dev, distributed, rehearsal):
    """BASELINE config 4: one mapping iteration over an 8-keyframe window at Replica resolution (1200x680), the
    keyframes sharded over the ranks, Gaussians replicated, one gradient all-reduce per iteration
    (monogs_amd.mapping.WindowMapper = /root/reference/utils/slam_mapper.py:244-500).  Strong scaling: the window is
    fixed, so at N ranks each renders 8/N keyframes.  Synthetic Replica-like sequence (the dataset is not available)."""
    import torch
    import torch.distributed as dist

    from monogs_amd import rasterizer as _rast
    from monogs_amd.gaussian_map import GaussianMap
    from monogs_amd.mapping import WindowMapper
    from monogs_amd.slam_harness import make_sequence

    frames, intr = make_sequence(args.window, "replica", n_gaussians=150_000, device=str(dev))
    bg = torch.zeros(3, device=dev)
    gmap = GaussianMap(str(dev))
    gmap.extend_from_frame(frames[0], intr, downsample=8, init=True, point_size=1.0)       # ~100 k Gaussians
    for vp in frames:
        vp.update_RT(vp.R_gt.clone(), vp.T_gt.clone())
    mapper = WindowMapper(gmap, intr, bg, window_size=args.window, use_graph=not args.eager)
    mapper.map_surgery = False                   # fixed workload: no densification / opacity reset inside the timed region
    mapper.exchange = args.exchange
    P, H, W = len(gmap), intr.height, intr.width

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"c4 scene ready: P={P} {W}x{H}, window {args.window}, world {world}; warmup {args.warmup}"
        + ("" if args.eager else " (hipGraph-replayed iterations)"))
    # warm-up = the eager first iteration (capacity hints) + the capture + replays; the timed call then replays only
    if args.eager:
        mapper.optimize_map(frames, iters=1)             # exact path once: records the capacity hints
        _rast.set_sync_free(True)                        # then no per-render host read-back, as inside a replay
    mapper.optimize_map(frames, iters=max(args.warmup, mapper.min_graph_iters + 1))
    fence()
    t0 = time.perf_counter()
    mapper.optimize_map(frames, iters=args.steps)
    fence()
    dt = time.perf_counter() - t0
    if _rast.check_overflow():
        raise SystemExit("capacity overflow during the timed region")
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # exposed time of the exchanges (separate iterations: the measurement synchronises around each collective)
    mapper.time_comm, mapper.exposed_comm_s = True, 0.0
    n_prof = max(1, args.profile_steps)
    mapper.optimize_map(frames, iters=n_prof)
    comm_ms = mapper.exposed_comm_s / n_prof * 1e3
    mapper.sync_poses(frames)
    in_sync, comm = True, None
    if distributed:
        from monogs_amd.window import replicas_in_sync
        in_sync = replicas_in_sync(gmap.params())
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices_visible": torch.cuda.device_count(),
                "device": torch.cuda.get_device_name(dev)}
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        line = {
            "metric": "mapping-window fwd+bwd Mpix/s (C4)", "value": round(args.steps * args.window * H * W / 1e6 / dt, 2),
            "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4: mapping iteration over a {args.window}-keyframe window, {W}x{H} (Replica intrinsics), "
                                   f"{P} Gaussians, keyframes sharded k % {world}; per keyframe render (screen-space holder, radii, "
                                   f"n_touched) + get_loss_mapping + backward; per-keyframe densification statistics + MAX radii + "
                                   f"visibility bits; all-reduce ({args.exchange}) + all-gather; fused Adam + xyz lr schedule + pose steps; "
                                   + ("eager" if args.eager else "hipGraph-replayed")
                                   + (" [REHEARSAL: ranks share a device, gloo]" if rehearsal else ""),
                       "gaussians": P, "width": W, "height": H, "window": args.window,
                       "parallelism": f"keyframe-sharded x{world}"},
            "mapping_iters_per_s": round(args.steps / dt, 2), "exchange_exposed_ms": round(comm_ms, 4),
            "exchange_bytes": {"allreduce_sum": 4 * P * 16, "allgather_per_rank": 8 * ((P + 1) // 2 + ((args.window + world - 1) // world) * ((P + 63) // 64))},
            "replicas_in_sync": bool(in_sync), "mapper_stats": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in mapper.stats.items()},
            "roofline": None, "cpu_baseline": None, "comm": comm,
        }
        print(json.dumps(line), flush=True)
    if not in_sync:
        print(f"[bench] rank {rank}: the replicated maps DIVERGED across the ranks", file=sys.stderr, flush=True)
        raise SystemExit(3)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # not under torch.distributed.run yet: spawn the ranks BEFORE torch.cuda / HIP is touched in this process
        raise SystemExit(self_launch(args))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()          # (does not initialise the GPU)
    assert n_dev > 0, "bench.py needs a GPU"
    # Fewer devices than ranks (a one-GPU box): REHEARSAL of the multi-rank path -- ranks share devices and the
    # collectives run over gloo, staged through host memory (RCCL refuses two ranks on one device).  Flagged in the line.
    rehearsal = world > n_dev
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    distributed = "RANK" in os.environ and "MASTER_PORT" in os.environ     # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    if args.workload == "c4":
        return bench_c4(args, rank, world, dev, distributed, rehearsal)

    from monogs_amd import rasterizer as _rast
    from monogs_amd.camera import se3_exp
    from monogs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, collect_timing
    from monogs_amd.synthetic import make_scene, scene_settings
    from monogs_amd.window import GradBucket

    # the map is shared by all ranks (seed 2); every rank looks at it from its own keyframe pose
    is_c5 = args.gaussians == 2_000_000 and args.intrinsics == "davis_1080p"
    sc = make_scene(args.gaussians, args.intrinsics, seed=2)
    if rank:
        d = se3_exp(torch.tensor([0.02 * rank, -0.01 * rank, 0.0, 0.0, 0.004 * rank, 0.0]))
        T = torch.eye(4)
        T[:3, :3], T[:3, 3] = sc.R, sc.t
        T = d @ T
        sc = sc._replace(R=T[:3, :3].contiguous(), t=T[:3, 3].contiguous())
    st = scene_settings(sc, GaussianRasterizationSettings, device=dev)
    H, W = st.image_height, st.image_width
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
    xyz, rgb, opac, scaling, rot = (leaf(sc.means3D), leaf(sc.colors), leaf(sc.opacities), leaf(sc.scales),
                                    leaf(sc.rotations))
    params = [xyz, rgb, opac, scaling, rot]
    theta = torch.zeros(3, device=dev, requires_grad=True)
    rho = torch.zeros(3, device=dev, requires_grad=True)
    g_color, g_depth = sc.grad_color.to(dev), sc.grad_depth.to(dev)
    rasterizer = GaussianRasterizer(st)
    bucket = GradBucket(params) if (distributed and world > 1) else None
    state = {}

    def step(stats=False, reduce=True):
        for p in params:
            p.grad = None
        theta.grad = rho.grad = None
        means2D = torch.zeros_like(xyz, requires_grad=True)
        color, radii, depth, opacity, n_touched = rasterizer(
            means3D=xyz, means2D=means2D, opacities=opac, colors_precomp=rgb, scales=scaling,     # isotropic [P,1], as MonoGS's map
            rotations=rot, theta=theta, rho=rho)
        if stats is True:
            state["walk"] = _rast.debug_blend_stats(color)
        torch.autograd.backward([color, depth], [g_color, g_depth])
        if bucket is not None and reduce:
            if stats == "time_exchange":            # (un-timed diagnostic steps: how long the collective itself takes)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            bucket.pack()
            bucket.all_reduce()
            bucket.unpack()
            if stats == "time_exchange":
                e1.record()
                state.setdefault("exchange_events", []).append((e0, e1))
        state["radii"] = radii

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def live_ceilings():
        """The two ceilings the roofline record quotes, measured live in this run: the plain-FMA issue rate (8 waves per SIMD, every
        CU busy: mgs_debug_valu_ceiling) and a 512 MiB device-to-device copy (read + write counted).  Measured
        after the timed region (rank 0 only)."""
        from monogs_amd import _lib as _L
        lib = _L.load()
        buf = torch.empty(2048 * 256, device=dev)
        it_fma = 20000
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _L.check(lib.mgs_debug_valu_ceiling(buf.data_ptr(), it_fma, torch.cuda.current_stream().cuda_stream), "valu_ceiling")
            e1.record()
            torch.cuda.synchronize()
        fma = 8.0 * it_fma * 2048 * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        src = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy = 10 * 2 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst, buf
        torch.cuda.empty_cache()
        return fma, copy

    log(f"scene ready: P={args.gaussians} {W}x{H}; warmup {args.warmup}")
    step()                                   # first step always exact: it records the capacity hint
    sync_free = not args.exact_count
    _rast.set_sync_free(sync_free)           # steady state: device-side instance count, no host sync per forward
    # What this process reads WITHOUT the settle steps: the first `cold_steps` steps after the exact one, between two fences
    # (`value_cold` in the line; they also count as un-timed steps in front of the headline's timed region).
    cold_steps = min(20, max(0, args.settle_steps)) if args.settle_steps > 0 else 0
    dt_cold = None
    if cold_steps:
        fence()
        tc = time.perf_counter()
        for _ in range(cold_steps):
            step()
        fence()
        dt_cold = time.perf_counter() - tc
        if distributed:
            tt = torch.tensor([dt_cold], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_cold = float(tt.item())
    for _ in range(max(0, args.settle_steps - cold_steps) + max(0, args.warmup - 1)):
        step()
    fence()
    warmup_effective = 1 + max(cold_steps, args.settle_steps) + max(0, args.warmup - 1)       # every step that ran before the timed region
    log(f"warmup done ({args.settle_steps} settle steps, the first {cold_steps} of them timed as the cold figure, + {args.warmup} warm-up steps "
        f"= {warmup_effective} un-timed steps); timing", args.steps, "steps")
    # (short runs: an event per step, so that the line's reader can see how far from the steady state the run was -- the two
    #  blend kernels take ~40 steps of sustained load from process start to reach their speed, see --settle-steps)
    trace = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if (args.trace_region and args.steps <= 64 and rank == 0) else None
    # the two blend kernels, timed INSIDE the timed region (rank 0): the library records a caller's events right around their
    # launches in up to 8 evenly spaced steps (mgs_debug_set_blend_events: no sync, two event records per kernel: ~4 us of
    # stream time per probed step, i.e. < 0.2 % of a 20-step region; round 4 probed every step of a short run).  The
    # per-stage profile further down synchronises per call, and a device that idles between kernels runs the issue-bound
    # blend kernels ~8 % slower than back-to-back steps do -- this is the figure rocprofv3's average agrees with.
    from monogs_amd import _lib as _L
    hook, probes = _L.load().mgs_debug_set_blend_events, {}
    if rank == 0:
        for i in range(0, args.steps, max(1, -(-args.steps // 8))):
            probes[i] = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for e in probes[i]:
                e.record()                  # (creates the handle)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    if trace:
        trace[0].record()
    for i in range(args.steps):
        if i in probes:
            _L.check(hook(*[e.cuda_event for e in probes[i]]), "mgs_debug_set_blend_events")
        step()
        if i in probes:
            hook(None, None, None, None)
        if trace:
            trace[i + 1].record()            # (short runs only: an event record per step, ~1 us of stream time)
            state.setdefault("reserved", []).append(torch.cuda.memory_reserved())
    t_host = time.perf_counter() - t0        # (diagnostic: when the host had queued the last step)
    fence()
    dt = time.perf_counter() - t0
    if trace:
        log("timed region, device ms per step:", [round(trace[i].elapsed_time(trace[i + 1]), 3) for i in range(args.steps)])
        log("timed region, GiB reserved by the caching allocator after each step:", [round(v / 2**30, 2) for v in state.get("reserved", [])])
    log(f"timed region: host had queued the {args.steps} steps after {t_host * 1e3:.2f} ms, the device was done after {dt * 1e3:.2f} ms")
    region = None
    if probes:
        f_ms = sorted(p[0].elapsed_time(p[1]) for p in probes.values())
        b_ms = sorted(p[2].elapsed_time(p[3]) for p in probes.values())
        region = {"blend_fwd_ms": round(sum(f_ms) / len(f_ms), 4), "blend_bwd_ms": round(sum(b_ms) / len(b_ms), 4),
                  "blend_fwd_ms_median": round(f_ms[len(f_ms) // 2], 4), "blend_bwd_ms_median": round(b_ms[len(b_ms) // 2], 4),
                  "launches_timed": len(probes)}
        log("blend kernels inside the timed region (HIP events around the launches):", region)
    if sync_free and _rast.check_overflow():
        raise SystemExit("capacity overflow during the timed region: rerun with --exact-count")
    if args.trace_steps > 0 and rank == 0 and not distributed:
        fence()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.trace_steps + 1)]
        host = [time.perf_counter()]
        evs[0].record()
        for i in range(args.trace_steps):
            step()
            evs[i + 1].record()
            host.append(time.perf_counter())
        torch.cuda.synchronize()
        log("per-step device ms behind a fence:", [round(evs[i].elapsed_time(evs[i + 1]), 3) for i in range(args.trace_steps)])
        log("per-step host enqueue ms:", [round(1e3 * (host[i + 1] - host[i]), 3) for i in range(args.trace_steps)])
    _rast.set_sync_free(False)               # the per-stage profile below uses the exact path
    log(f"timed region: {dt / args.steps * 1e3:.3f} ms/step" + (" (capacity mode, no per-forward host sync)" if sync_free else ""))
    if distributed:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- N > 1: the exchange on its own (events around the collective, every rank runs it; rank 0 reports)
    exchange = None
    if bucket is not None:
        for _ in range(5):
            step(stats="time_exchange")
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in state["exchange_events"])
        nbytes = int(sum(p.numel() for p in params) * 4)
        t_ex = ms[len(ms) // 2] * 1e-3
        exchange = {"bytes": nbytes, "allreduce_ms": round(ms[len(ms) // 2], 4),
                    "allreduce_ms_min_max": [round(ms[0], 4), round(ms[-1], 4)],
                    # nccl-tests convention: algorithm bandwidth = bytes / time, bus bandwidth = x 2 (N - 1) / N (what each link
                    # of a ring carries); xGMI: 7 links x ~153 GB/s per GPU
                    "algbw_gbs": round(nbytes / t_ex / 1e9, 2), "busbw_gbs": round(nbytes / t_ex / 1e9 * 2 * (world - 1) / world, 2),
                    "collectives_per_step": bucket.last_collectives,
                    "note": "median of 5 un-timed steps, HIP events around pack + collective + unpack on the launch stream; "
                            "ms_per_step contains it in full (the gradients are complete only when the backward ends)"}
        # every rank must hold the same reduced gradients, bit for bit (what keeps the replicated maps identical with no
        # parameter broadcast): a 64-bit checksum per tensor, MIN / MAX over the ranks
        from monogs_amd.window import replicas_in_sync
        in_sync = bool(replicas_in_sync([p.grad for p in params]))
        exchange["replicas_in_sync"] = in_sync
        if not in_sync:       # how far apart: max |g - g of rank 0| over the ranks (a collective that sums in a rank-dependent order
            worst = 0.0       # would show up here as a last-bit difference, a broken exchange as something large)
            for p in params:
                ref = p.grad.detach().clone()
                if rehearsal:
                    h = ref.cpu(); dist.broadcast(h, src=0); ref = h.to(dev)
                else:
                    dist.broadcast(ref, src=0)
                d = (p.grad - ref).abs().max().reshape(1)
                if rehearsal:
                    h = d.cpu(); dist.all_reduce(h, op=dist.ReduceOp.MAX); d = h
                else:
                    dist.all_reduce(d, op=dist.ReduceOp.MAX)
                worst = max(worst, float(d.item()) / max(float(ref.abs().max().item()), 1e-30))
            exchange["max_relative_difference_between_ranks"] = worst

    # ---- per-kernel time, live, with HIP events on the launch stream (separate from the timed region)
    roof = None
    stages = {}
    # every rank runs the same profile steps (they contain the collective); only rank 0 records stage times
    import contextlib
    with (collect_timing() if rank == 0 else contextlib.nullcontext([])) as sink:
        for _ in range(max(1, args.profile_steps)):
            step()
        torch.cuda.synchronize()
    if rank == 0:
        step(stats=True, reduce=False)       # (un-timed, rank 0 only: no collective) what the backward walks
        walk = state.get("walk")
        fw = [d for d in sink if d["kind"] == "forward"]
        bw = [d for d in sink if d["kind"] == "backward"]
        # median over the profiled steps (the first one pays for the events' creation: its duplicate stage read 3x once)
        avg = lambda rows, k: sorted(r[k] for r in rows)[len(rows) // 2] if rows else 0.0  # noqa: E731
        for k in ("preprocess_ms", "depth_sort_ms", "scan_ms", "duplicate_ms", "sort_ms", "ranges_ms", "blend_fwd_ms"):
            stages[k] = round(avg(fw, k), 4)
        for k in ("blend_bwd_ms", "geom_bwd_ms"):
            stages[k] = round(avg(bw, k), 4)
        R = fw[0]["num_rendered"]
        Pv = int((state["radii"] > 0).sum().item())
        HW = H * W
        P = args.gaussians
        # algorithmic bytes (SURVEY.md section 8d / BASELINE.md section 4)
        b_fwd = 44 * R + 28 * HW + 4 * Pv
        b_bwd = 44 * R + 24 * HW + 40 * Pv
        # the dominant kernel's average launch duration: inside the timed region when the hook ran (rank 0), else the profile steps'
        bwd_ms = region["blend_bwd_ms"] if region else stages["blend_bwd_ms"]
        fwd_ms = region["blend_fwd_ms"] if region else stages["blend_fwd_ms"]
        ach = b_bwd / (bwd_ms * 1e-3) / 1e9
        both = (b_fwd + b_bwd) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9
        # PMC-derived numbers are attached only when they were collected on exactly these kernel sources (and on C5)
        traffic = valu = None
        tj, vj = stamped_profile("traffic.json"), stamped_profile("pmc_valu.json")
        if is_c5 and tj:
            traffic = tj.get("blend_backward_bytes_per_launch")
        # ---- VALU view of the dominant kernel.  The data sheet's FP32 vector rate is one wave64 v_fma_f32 per 2 cycles per
        # SIMD (MI355X_MICROARCH.md: "v_fma_f32 (wave64): 2 cyc"; 157.3 TFLOP/s / 128 flop): 1024 x 2.4e9 / 2 = 1228.8 G
        # wave-inst/s.  What the chip sustains on plain FMAs under this load is measured LIVE below (it does not hold 2.4 GHz).
        # Neither is the bound that binds: the kernel's instructions are not all plain -- the issue model prices the static mix
        # of its hot loop (profiles/isa_mix.json) with the measured per-class issue times (profiles/valu_costs.json).
        fma_ginst, copy_gbs = live_ceilings()
        surv = max(1, walk["survivors"])
        t_bwd = bwd_ms * 1e-3
        ns_meas = t_bwd * 1e9 * 1024 / surv                 # SIMD-time per survivor: the launch's survivors spread over 1024 SIMDs
        valu = {"peak_ginst_s": 1228.8, "measured_fma_ginst_s": round(fma_ginst, 1),
                "survivors_per_launch": walk["survivors"], "active_survivors_per_launch": walk["active_survivors"],
                "ns_of_simd_time_per_survivor": round(ns_meas, 2)}
        mj, cj = stamped_profile("isa_mix.json"), None
        try:
            cj = json.load(open(os.path.join(ROOT, "profiles", "valu_costs.json")))
        except Exception:
            pass
        if mj and cj and mj.get("blend_backward_s_kernel<false>"):
            k = mj["blend_backward_s_kernel<false>"]
            per_s, per_b, bs = k["per_survivor"], k["per_batch"], float(mj.get("batch_size", 4))
            classes = ("valu_plain", "valu_trans", "valu_dpp", "valu_cndmask", "valu_lane", "valu_permlane_swap")
            cnt = {c: per_s.get(c, 0) + per_b.get(c, 0) / bs for c in classes}
            ns_model = sum(cnt[c] * float(cj[c]) for c in classes)
            valu["issue_model"] = {
                "valu_insts_per_survivor": round(sum(cnt.values()), 2), "mix_per_survivor": {c: round(v, 2) for c, v in cnt.items() if v},
                "salu_insts_per_survivor": round(per_s.get("salu", 0) + per_s.get("branch", 0) + (per_b.get("salu", 0) + per_b.get("branch", 0)) / bs, 2),
                "ns_per_survivor_modelled": round(ns_model, 2), "frac_of_issue_model": round(ns_model / ns_meas, 4),
                "costs_ns": {c: cj[c] for c in classes},
                "source": "profiles/isa_mix.json (static mix of the hot loop, stamped) x profiles/valu_costs.json (tools/ubench/valu_rate.hip)"}
            # the second pipe (round 5): the compute unit's ONE scalar ALU serves four SIMDs.  Every scalar-pipe instruction of the
            # hot loop (SALU, branches, scalar loads, waits / nops) is priced at what tools/ubench/valu_rate.hip measures a scalar
            # instruction to ADD to a vector stream at the kernels' own ratio (one scalar per two vector instructions).
            sc_cls = ("salu", "branch", "smem", "wait_nop")
            n_sc = sum(per_s.get(c, 0) for c in sc_cls) + sum(per_b.get(c, 0) for c in sc_cls) / bs
            if cj.get("salu_behind_2_valu"):
                ns_two = ns_model + n_sc * float(cj["salu_behind_2_valu"])
                valu["issue_model"].update({
                    "scalar_pipe_insts_per_survivor": round(n_sc, 2), "scalar_ns_each_alone": cj.get("salu_alone"),
                    "scalar_ns_each_behind_two_vector": cj["salu_behind_2_valu"],
                    "ns_per_survivor_modelled_two_pipes": round(ns_two, 2), "frac_of_two_pipe_model": round(ns_two / ns_meas, 4),
                    "two_pipe_note": "vector classes x their issue times + scalar-pipe instructions x the time one adds behind two v_fma "
                                     "(both priced in an all-FMA loop that clocks 1.7-1.9 GHz; the blend kernels run at 2.2 GHz, "
                                     "profiles/r05_kernel_clocks.txt, so the model is an upper estimate of the steady state: a fraction "
                                     "above 1 says the loop issues faster than the priced parts add up to).  The measured time also holds "
                                     "the launch's ramp and tail (tools/tail_probe.py: a fixed ~52 us per backward launch, ~15 % at C5)"})
        if is_c5 and vj and vj.get("blend_backward_s_kernel"):
            insts = float(vj["blend_backward_s_kernel"]["SQ_INSTS_VALU"])
            valu.update({"valu_wave_insts_per_launch": insts, "insts_per_survivor": round(insts / surv, 2),
                         "achieved_ginst_s": round(insts / t_bwd / 1e9, 1),
                         "frac_of_spec_valu_peak": round(insts / t_bwd / 1e9 / 1228.8, 4),
                         "frac_of_measured_fma_rate": round(insts / t_bwd / 1e9 / fma_ginst, 4),
                         "pmc_source": "profiles/pmc_valu.json (rocprofv3 --pmc, stamped with the kernel-source hash)"})
        # all stages: SURVEY.md 8d per-stage algorithmic bytes (one ideal sort pass each for the two sorts)
        b_all = (44 * P + 8 * P + 52 * Pv) + 8 * P + (16 * Pv + 12 * R) + 24 * R + (8 * R + 8 * ((W + 15) // 16) * ((H + 15) // 16)) \
            + b_fwd + b_bwd + (44 * P + 40 * Pv + 68 * Pv + 24)
        t_all = sum(stages.values())
        roof = {"bound": "hbm", "kernel": "blend_backward_s_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "valu": valu,
                "csrc_sha256": csrc_hash(),
                "algorithmic_bytes": b_bwd, "avg_ms": bwd_ms,
                "avg_ms_source": ("HIP events around the kernel's launches inside the timed region, mean of %d launches "
                                  "(mgs_debug_set_blend_events)" % region["launches_timed"]) if region else
                                 "HIP events between the stages of the profile steps (outside the timed region)",
                "avg_ms_profile_steps": stages["blend_bwd_ms"],
                "blend_fwd_bwd": {"achieved": round(both, 2), "frac": round(both / HBM_PEAK_GBS, 5),
                                  "algorithmic_bytes": b_fwd + b_bwd,
                                  "avg_ms": round(fwd_ms + bwd_ms, 4)},
                "all_stages": {"achieved": round(b_all / (t_all * 1e-3) / 1e9, 2), "algorithmic_bytes": b_all,
                               "frac": round(b_all / (t_all * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "avg_ms": round(t_all, 4),
                               "frac_over_ms_per_step": round(b_all / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5)},
                "copy_ceiling": {"measured_gbs": round(copy_gbs, 1), "frac_of_copy": round(ach / copy_gbs, 5),
                                 "blend_fwd_bwd_frac_of_copy": round(both / copy_gbs, 5)},
                "num_rendered": R, "visible": Pv, "backward_walk": walk}

    log("stages_ms", stages)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle) ...")
        cpu = cpu_baseline()
        log("cpu baseline", cpu)

    # ---- the second half of the metric: tracking + mapping rates on synthetic RGB-D stand-ins for BASELINE configs 3 / 4
    #      (the TUM / Replica sequences are not on the box): an OPAQUE box room with furniture, ray-cast analytically
    #      (monogs_amd.slam_harness.make_room_sequence), so that the fitted map survives the reference's 0.7 opacity pruning and
    #      the whole of optimize_map runs inside the timed loops: densify_and_prune every 150 iterations, opacity resets,
    #      covisibility pruning after every keyframe (/root/reference/utils/slam_mapper.py:408-451,462-480), with the fork's own
    #      new-Gaussian recipe (1/32 and 1/64 of the pixels, scale^2 = dist2 x min(0.05, 0.01 x median depth),
    #      gaussian_model.py:166-178), its learning rates and xyz schedule, and the Scharr gradient mask of camera_utils.py:185-216.
    #      Mapping = monogs_amd.mapping.WindowMapper, the SAME optimize_map / initialize_map the sharded C4 window runs, replayed
    #      from hipGraphs; every map-size change (surgery, new keyframe) costs a re-capture, which is INSIDE the rates.
    slam = None
    if rank == 0 and world == 1 and not args.no_slam:
        try:
            log("slam harness (tracking + mapping rates, map surgery on) ...")
            del xyz, rgb, opac, scaling, rot, params, g_color, g_depth
            torch.cuda.empty_cache()
            from monogs_amd.slam_harness import run_slam
            keys = ("tracking_fps", "tracking_iters_per_s", "mapping_iters_per_s", "mapping_kf_per_s",
                    "tracking_steady_iters_per_s", "mapping_steady_iters_per_s", "mapping_keyframe_iters_per_s",
                    "kf_extend_ms", "ate_rmse_m", "gaussians", "width", "height", "frames", "config", "window_sizes",
                    "mapping_replays", "mapping_eager_iters", "mapping_captures", "mapping_capture_s", "map_surgery", "surgery",
                    "eager_tracking")
            # (exclusive_device: the tracking graphs do NOT claim MGS_FLAG_EXCLUSIVE_DEVICE -- MonoGS's tracker shares the GPU with a
            #  mapper and a viewer process; the ~2 us per launch it would save were inside the noise anyway)
            common = dict(scene="room", reference_densify=True, map_surgery=True, reference_lrs=True, graph_tracking=True,
                          graph_mapping=True, exclusive_device=False)
            standin = ("opaque box room with furniture, ray-cast analytically (closed-form colour + z-depth per pixel), hand-held-like "
                       "path ~1 cm / 0.3 deg per frame; map surgery ON (densify_and_prune / opacity reset / covisibility prune on the "
                       "reference's schedule), fork's new-Gaussian recipe (1/32, 1/64, point-size rule), reference learning rates + "
                       "xyz schedule, Scharr gradient mask; hipGraph-replayed tracking and mapping iterations, graph re-captures "
                       "inside the rates; remaining stand-in deviations: no sensor noise, no exposure change, keyframes every "
                       "kf_interval frames (no overlap test), oldest-but-first keyframe leaves a full window")

            def block(r, what):
                d = {k: (round(v, 6 if k == "ate_rmse_m" else 3) if isinstance(v, float) else v) for k, v in r.items() if k in keys}
                d["max_window_reached"] = max(d["window_sizes"]) if d.get("window_sizes") else 0
                d["exclusive_device"] = bool(common["exclusive_device"])
                if isinstance(d.get("surgery"), dict):
                    d["surgery"] = {k: v for k, v in d["surgery"].items() if k != "log"}
                d["workload"] = what
                return d
            # (a short run first: module loading, lazy allocations and the first graph instantiation are one-time costs
            #  of the process -- 150 ms of them sat in the first keyframe of a six-frame run)
            run_slam(n_frames=2, intrinsics="fr3_office", tracking_itr_num=20, mapping_itr_num=20, window_size=8,
                     kf_interval=1, init_itr_num=20, **common)
            # YAML values of /root/reference/configs/mono/tum/base_config.yaml:19-36 (init 1050, tracking 100, mapping 150,
            # window 8, kf 5); the eager unmodified-caller probe runs against this run's final map
            r = run_slam(n_frames=41, intrinsics="fr3_office", tracking_itr_num=100, mapping_itr_num=150, window_size=8,
                         kf_interval=5, init_itr_num=1050, eager_probe=200, **common)
            slam = block(r, "stand-in for C3 (TUM fr3_office intrinsics, 640x480), YAML run values: " + standin)
            log("slam", slam)
            # /root/reference/configs/rgbd/replica/base_config.yaml:33-48 (init 1050, window 10, kf 4) at 1200x680
            r2 = run_slam(n_frames=41, intrinsics="replica", tracking_itr_num=100, mapping_itr_num=150, window_size=10,
                          kf_interval=4, init_itr_num=1050, **common)
            slam["replica_like"] = block(r2, "stand-in for C4's sequence (Replica intrinsics, 1200x680), YAML run values; same room, same recipe")
            log("slam replica-like", slam["replica_like"])
            # the values the fork hard-codes over its YAML: tracking 100, every frame a keyframe, init 1050, 300 iterations
            # per keyframe, window 30 (/root/reference/utils/slam_tracker.py:70-72, utils/slam_mapper.py:64-89,660-662, slam.py:75)
            r3 = run_slam(n_frames=32, intrinsics="fr3_office", tracking_itr_num=100, mapping_itr_num=300, window_size=30,
                          kf_interval=1, init_itr_num=1050, **common)
            slam["fork_hardcoded"] = block(r3, "the same TUM-like stand-in with the fork's hard-coded run configuration "
                                               "(init 1050, 300 iterations per keyframe, window 30, every frame a keyframe)")
            log("slam fork", slam["fork_hardcoded"])
            # the unmodified-caller probe once more, against a map of the size the reference's runs END with before pruning bites
            # (~39 k Gaussians: the semi-transparent cloud of rounds 1-3, no surgery) -- the five flavours at 10 k above, at 39 k here
            r4 = run_slam(n_frames=3, intrinsics="fr3_office", tracking_itr_num=20, mapping_itr_num=20, window_size=8, kf_interval=1,
                          init_itr_num=50, n_gaussians=60000, scene="cloud", eager_probe=100)
            slam["eager_tracking_39k"] = r4.get("eager_tracking")
            log("slam eager 39k", slam["eager_tracking_39k"])
        except Exception as e:          # never lose the headline line to the auxiliary measurement
            import traceback
            traceback.print_exc()
            slam = {"error": repr(e)[:300]}

    comm = None
    if distributed:
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices_visible": n_dev,
                "device": torch.cuda.get_device_name(dev)}
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        mpix = world * args.steps * H * W / 1e6 / dt
        line = {
            "metric": "rasteriser fwd+bwd Mpix/s @1080p", "value": round(mpix, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # every step this process ran before the timed region: the exact first step, the settle steps, the warm-up steps
            "warmup_effective": warmup_effective,
            # the same workload timed WITHOUT the settle steps: the first `cold_steps` steps after the exact one (None: --settle-steps 0,
            # then `value` itself is that figure)
            "value_cold": (round(world * W * H * cold_steps / dt_cold / 1e6, 2) if dt_cold else None),
            "ms_per_step_cold": (round(dt_cold / cold_steps * 1e3, 4) if dt_cold else None), "cold_steps": cold_steps,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'C5' if is_c5 else 'custom'}: {args.gaussians} Gaussians, {W}x{H}, fwd+bwd, seeded synthetic map "
                                   f"(SURVEY.md 8d), one keyframe per GPU",
                       "gaussians": args.gaussians, "width": W, "height": H, "settle_steps": args.settle_steps,
                       "instance_count": "device-side (capacity mode, overflow checked)" if sync_free else "host read-back per forward",
                       "parallelism": f"keyframe-per-gpu x{world}" + (f" + RCCL all-reduce of 12 floats/Gaussian in {bucket.last_collectives} collective(s)" if bucket is not None else "")
                       + (" [REHEARSAL: ranks share a device, collectives over gloo]" if rehearsal else "")},
            "stages_ms": stages,
            "stages_note": "HIP events between the stages, in profile steps outside the timed region (median); every profiled call "
                           "synchronises, and behind an idle gap the issue-bound blend kernels run ~8 % slower than in back-to-back "
                           "steps, so the stages add up to more than ms_per_step; blend_in_timed_region has the two blend kernels as "
                           "the timed region runs them (what roofline is computed from)",
            "blend_in_timed_region": region,
            "roofline": roof, "cpu_baseline": cpu, "slam": slam,
        }
        if exchange is not None:
            line["exchange"] = exchange
            line["replicas_in_sync"] = exchange["replicas_in_sync"]
        if comm is not None:
            line["comm"] = comm
        print(json.dumps(line), flush=True)
    if exchange is not None and not exchange["replicas_in_sync"]:
        # (every rank computed the same verdict from the same MIN / MAX reductions: all of them leave non-zero)
        print(f"[bench] rank {rank}: the ranks hold DIFFERENT reduced gradients after the all-reduce", file=sys.stderr, flush=True)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
