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
    ap.add_argument("--gaussians", type=int, default=60000)
    ap.add_argument("--graph", action="store_true", help="capture the tracking iteration in a hipGraph (capacity mode)")
    ap.add_argument("--eager-mapping", action="store_true", help="with --graph: capture tracking only")
    ap.add_argument("--torch-pose", action="store_true", help="torch.optim.Adam + Python retraction instead of mgs_pose_step")
    ap.add_argument("--torch-losses", action="store_true", help="use the plain PyTorch losses instead of the fused HIP ones")
    ap.add_argument("--serial-kf", action="store_true", help="render the window's keyframes one after the other (default with --graph: a stream each)")
    ap.add_argument("--lookahead", type=int, default=1, choices=[0, 1],
                    help="with --graph: read the convergence flag of tracking iteration n-1 while n runs")
    a = ap.parse_args()
    from monogs_amd.slam_harness import run_slam
    loss_module = None
    if a.torch_losses:          # measurement tool only: the PyTorch mirror lives with the test infrastructure
        from oracle import slam_losses as loss_module
    cfg = dict(CONFIGS[a.config])
    if a.mapping_iters is not None:
        cfg["mapping_itr_num"] = a.mapping_iters
    out = run_slam(n_frames=a.frames, init_itr_num=a.init_iters, n_gaussians=a.gaussians,
                   fused_losses_on=not a.torch_losses, fused_pose_on=not a.torch_pose, graph_tracking=a.graph, graph_mapping=a.graph and not a.eager_mapping, track_lookahead=a.lookahead, loss_module=loss_module, parallel_keyframes=False if a.serial_kf else None,
                   log=lambda s: print("[slam]", s, file=sys.stderr, flush=True), **cfg)
    out["workload"] = f"synthetic {a.config}-like sequence, {a.frames} frames"
    for k in ("poses", "camera_centers", "camera_centers_gt"):      # tensors: not JSON
        out.pop(k, None)
    print(json.dumps(out))
