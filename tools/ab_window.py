#!/usr/bin/env python3
"""Mapping-window iterations per second on a FIXED map (no SLAM run around it, so two runs do the same work): an
8-keyframe window of the synthetic TUM-like sequence (640x480, ~39 k Gaussians, ~415 k instances per keyframe) replayed
from hipGraphs, the keyframes on a stream each.  For A/B runs of knobs that matter under concurrency
(MGS_DEBUG_OPTIONS="radix_scanned=1" python tools/ab_window.py)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from monogs_amd import rasterizer as _rast  # noqa: E402
from monogs_amd.gaussian_map import GaussianMap  # noqa: E402
from monogs_amd.mapping import WindowMapper  # noqa: E402
from monogs_amd.slam_harness import make_sequence  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--intrinsics", default="fr3_office")
ap.add_argument("--window", type=int, default=8)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
frames, intr = make_sequence(a.window, a.intrinsics, n_gaussians=60000, device=str(dev))
gmap = GaussianMap(str(dev))
gmap.extend_from_frame(frames[0], intr, downsample=8, init=True, point_size=1.0)
for vp in frames:
    vp.update_RT(vp.R_gt.clone(), vp.T_gt.clone())
mapper = WindowMapper(gmap, intr, torch.zeros(3, device=dev), window_size=a.window, use_graph=True)
mapper.map_surgery = False
mapper.optimize_map(frames, iters=mapper.min_graph_iters + 12)
rates = []
for _ in range(a.reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mapper.optimize_map(frames, iters=a.steps)
    torch.cuda.synchronize()
    rates.append(a.steps / (time.perf_counter() - t0))
if _rast.check_overflow():
    raise SystemExit("capacity overflow")
print(json.dumps({"options": os.environ.get("MGS_DEBUG_OPTIONS", ""), "gaussians": len(gmap), "window": a.window,
                  "mapping_iters_per_s": [round(r, 1) for r in rates]}))
