#!/usr/bin/env python3
"""Rasteriser fwd+bwd throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step is one forward+backward render of BASELINE config 5 (2M Gaussians, 1920x1080, synthetic):
preprocess, binning, blend, blend backward, per-Gaussian backward with the pose Jacobian.  With
N > 1 (launched by torch.distributed.run, one rank per GPU) every rank renders its own keyframe of
the mapping window against the same Gaussians and the ranks all-reduce the Gaussian gradients over
RCCL (weak scaling: one 1080p keyframe per GPU).  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.
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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=2_000_000)
    ap.add_argument("--intrinsics", default="davis_1080p")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-slam", action="store_true", help="skip the short tracking+mapping run (N=1 only)")
    ap.add_argument("--exact-count", action="store_true",
                    help="read the instance count back every forward (upstream behaviour) instead of capacity mode")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps timed per stage with HIP events")
    return ap.parse_args()


def cpu_baseline():
    """The oracle (float32 PyTorch-CPU autograd rasteriser) on a density-preserving crop of the
    workload: same focal length and generator, 1/4 of the pixels and of the Gaussians (~15 s of CPU work)."""
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
    t0 = time.perf_counter()
    rasterize_autograd(inp, st, sc.grad_color, sc.grad_depth, dtype=torch.float32)
    dt = time.perf_counter() - t0
    return {"value": round(intr["W"] * intr["H"] / 1e6 / dt, 5), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": f"500k Gaussians, 960x540 crop of the 1080p workload (same focal length, same per-pixel "
                      f"density), one fwd+bwd, float32, {dt:.1f} s"}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = "RANK" in os.environ and "MASTER_PORT" in os.environ     # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

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
    bucket = GradBucket(params) if distributed else None
    state = {}

    def step():
        for p in params:
            p.grad = None
        theta.grad = rho.grad = None
        means2D = torch.zeros_like(xyz, requires_grad=True)
        color, radii, depth, opacity, n_touched = rasterizer(
            means3D=xyz, means2D=means2D, opacities=opac, colors_precomp=rgb, scales=scaling,     # isotropic [P,1], as MonoGS's map
            rotations=rot, theta=theta, rho=rho)
        torch.autograd.backward([color, depth], [g_color, g_depth])
        if bucket is not None:
            bucket.pack()
            bucket.all_reduce()
            bucket.unpack()
        state["radii"] = radii

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"scene ready: P={args.gaussians} {W}x{H}; warmup {args.warmup}")
    from monogs_amd import rasterizer as _rast
    step()                                   # first step always exact: it records the capacity hint
    sync_free = not args.exact_count
    _rast.set_sync_free(sync_free)           # steady state: device-side instance count, no host sync per forward
    for _ in range(max(0, args.warmup - 1)):
        step()
    fence()
    log("warmup done; timing", args.steps, "steps")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if sync_free and _rast.check_overflow():
        raise SystemExit("capacity overflow during the timed region: rerun with --exact-count")
    _rast.set_sync_free(False)               # the per-stage profile below uses the exact path
    log(f"timed region: {dt / args.steps * 1e3:.3f} ms/step" + (" (capacity mode, no per-forward host sync)" if sync_free else ""))
    if distributed:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

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
        fw = [d for d in sink if d["kind"] == "forward"]
        bw = [d for d in sink if d["kind"] == "backward"]
        avg = lambda rows, k: sum(r[k] for r in rows) / max(1, len(rows))  # noqa: E731
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
        ach = b_bwd / (stages["blend_bwd_ms"] * 1e-3) / 1e9
        both = (b_fwd + b_bwd) / ((stages["blend_fwd_ms"] + stages["blend_bwd_ms"]) * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if is_c5 and os.path.exists(tpath):          # the PMC passes were collected on the C5 workload only
            try:
                traffic = json.load(open(tpath)).get("blend_backward_bytes_per_launch")
            except Exception:
                traffic = None
        # measured device-copy ceiling from the same run: 512 MiB device-to-device copy, read + write counted
        src = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
        # all stages: SURVEY.md 8d per-stage algorithmic bytes (one ideal sort pass each for the two sorts)
        b_all = (44 * P + 8 * P + 52 * Pv) + 8 * P + (16 * Pv + 12 * R) + 24 * R + (8 * R + 8 * ((W + 15) // 16) * ((H + 15) // 16)) \
            + b_fwd + b_bwd + (44 * P + 40 * Pv + 68 * Pv + 24)
        t_all = sum(stages.values())
        roof = {"bound": "hbm", "kernel": "blend_backward_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes": b_bwd, "avg_ms": stages["blend_bwd_ms"],
                "blend_fwd_bwd": {"achieved": round(both, 2), "frac": round(both / HBM_PEAK_GBS, 5),
                                  "algorithmic_bytes": b_fwd + b_bwd,
                                  "avg_ms": round(stages["blend_fwd_ms"] + stages["blend_bwd_ms"], 4)},
                "all_stages": {"achieved": round(b_all / (t_all * 1e-3) / 1e9, 2), "algorithmic_bytes": b_all,
                               "frac": round(b_all / (t_all * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "avg_ms": round(t_all, 4)},
                "copy_ceiling": {"measured_gbs": round(copy_gbs, 1), "frac_of_copy": round(ach / copy_gbs, 5),
                                 "blend_fwd_bwd_frac_of_copy": round(both / copy_gbs, 5)},
                "num_rendered": R, "visible": Pv}

    log("stages_ms", stages)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle) ...")
        cpu = cpu_baseline()
        log("cpu baseline", cpu)

    # ---- the second half of the metric: tracking + mapping rates on a short synthetic TUM-like sequence
    slam = None
    if rank == 0 and world == 1 and not args.no_slam:
        try:
            log("slam harness (tracking + mapping rates) ...")
            del xyz, rgb, opac, scaling, rot, params, g_color, g_depth
            torch.cuda.empty_cache()
            from monogs_amd.slam_harness import run_slam
            r = run_slam(n_frames=6, intrinsics="fr3_office", tracking_itr_num=100, mapping_itr_num=150, window_size=8,
                         kf_interval=5, init_itr_num=150, graph_tracking=True, graph_mapping=True)
            slam = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()
                    if k in ("tracking_fps", "tracking_iters_per_s", "mapping_iters_per_s", "mapping_kf_per_s",
                             "tracking_steady_iters_per_s", "mapping_steady_iters_per_s", "kf_extend_ms",
                             "ate_rmse_m", "gaussians", "width", "height", "frames", "config", "graph_tracking", "graph_mapping")}
            slam["workload"] = "synthetic TUM-like sequence (fr3_office intrinsics), hipGraph-captured tracking and mapping iterations"
            log("slam", slam)
        except Exception as e:          # never lose the headline line to the auxiliary measurement
            slam = {"error": repr(e)[:200]}

    if distributed:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        mpix = world * args.steps * H * W / 1e6 / dt
        line = {
            "metric": "rasteriser fwd+bwd Mpix/s @1080p", "value": round(mpix, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'C5' if is_c5 else 'custom'}: {args.gaussians} Gaussians, {W}x{H}, fwd+bwd, seeded synthetic map "
                                   f"(SURVEY.md 8d), one keyframe per GPU",
                       "gaussians": args.gaussians, "width": W, "height": H,
                       "instance_count": "device-side (capacity mode, overflow checked)" if sync_free else "host read-back per forward",
                       "parallelism": f"keyframe-per-gpu x{world}" + (" + RCCL all-reduce of 12 floats/Gaussian" if world > 1 else "")},
            "stages_ms": stages, "roofline": roof, "cpu_baseline": cpu, "slam": slam,
        }
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
