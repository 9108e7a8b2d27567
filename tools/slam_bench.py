#!/usr/bin/env python3
"""Tracking + mapping FPS of the two SLAM hot loops on a synthetic sequence (BASELINE configs 3-4
stand-in; the TUM / Replica sequences are not available offline).  Prints one JSON line.
  python tools/slam_bench.py --config tum      # 640x480, tracking 100 / mapping 150 / window 8 / kf 5
  python tools/slam_bench.py --config replica  # 1200x680, tracking 100 / mapping 150 / window 10 / kf 4
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS = {
    # /root/reference/configs/mono/tum/base_config.yaml:24-33
    "tum": dict(intrinsics="fr3_office", tracking_itr_num=100, mapping_itr_num=150, window_size=8, kf_interval=5),
    # /root/reference/configs/rgbd/replica/base_config.yaml:39-48
    "replica": dict(intrinsics="replica", tracking_itr_num=100, mapping_itr_num=150, window_size=10, kf_interval=4),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tum", choices=list(CONFIGS))
    ap.add_argument("--frames", type=int, default=11)
    ap.add_argument("--init-iters", type=int, default=300)
    ap.add_argument("--mapping-iters", type=int, default=None)
    ap.add_argument("--tracking-iters", type=int, default=None)
    ap.add_argument("--window", type=int, default=None)
    ap.add_argument("--kf-interval", type=int, default=None)
    ap.add_argument("--gaussians", type=int, default=60000)
    ap.add_argument("--graph", action="store_true", help="replay the tracking and the mapping iteration from hipGraphs (capacity mode)")
    ap.add_argument("--eager-mapping", action="store_true", help="with --graph: capture tracking only")
    ap.add_argument("--surgery", action="store_true", help="densify_and_prune / opacity resets / covisibility pruning on the reference's schedule")
    ap.add_argument("--reference-lrs", action="store_true", help="the reference's learning rates + xyz schedule")
    ap.add_argument("--room", action="store_true", help="opaque box-room sequence (ray-cast) instead of the semi-transparent cloud")
    ap.add_argument("--reference-densify", action="store_true",
                    help="new Gaussians as the fork hard-codes them: 1/32 (init) and 1/64 of the pixels, scale^2 = dist2 x min(0.05, 0.01 x median depth)")
    ap.add_argument("--eager-probe", type=int, default=0, help="after the run: N iterations of the unmodified eager tracking loop, timed")
    ap.add_argument("--fork", action="store_true",
                    help="the values the fork hard-codes over its YAML (/root/reference/utils/slam_tracker.py:70-72, "
                         "utils/slam_mapper.py:64-89,660-662, slam.py:75): tracking 100, every frame a keyframe, init 1050, "
                         "300 iterations per keyframe, window 30")
    ap.add_argument("--lookahead", type=int, default=1, choices=[0, 1],
                    help="with --graph: read the convergence flag of tracking iteration n-1 while n runs")
    a = ap.parse_args()
    from monogs_amd.slam_harness import run_slam
    cfg = dict(CONFIGS[a.config])
    init_iters = a.init_iters
    if a.fork:
        cfg.update(tracking_itr_num=100, mapping_itr_num=300, window_size=30, kf_interval=1)
        init_iters = 1050
    for k, v in (("mapping_itr_num", a.mapping_iters), ("tracking_itr_num", a.tracking_iters), ("window_size", a.window),
                 ("kf_interval", a.kf_interval)):
        if v is not None:
            cfg[k] = v
    out = run_slam(n_frames=a.frames, init_itr_num=init_iters, n_gaussians=a.gaussians, graph_tracking=a.graph,
                   graph_mapping=a.graph and not a.eager_mapping, track_lookahead=a.lookahead, map_surgery=a.surgery,
                   reference_lrs=a.reference_lrs, scene="room" if a.room else "cloud", reference_densify=a.reference_densify,
                   eager_probe=a.eager_probe, log=lambda s: print("[slam]", s, file=sys.stderr, flush=True), **cfg)
    out["workload"] = f"synthetic {a.config}-like sequence, {a.frames} frames" + (" (fork's hard-coded run configuration)" if a.fork else "")
    for k in ("poses", "camera_centers", "camera_centers_gt"):      # tensors: not JSON
        out.pop(k, None)
    print(json.dumps(out))
