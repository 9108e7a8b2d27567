#!/usr/bin/env python3
"""Where does the HOST time of an eager render + backward go?  (cProfile over a few hundred small iterations)"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from monogs_amd.slam_harness import Intrinsics, Viewpoint
from monogs_amd.renderer import render
from monogs_amd.synthetic import make_scene
from monogs_amd import fused_losses

dev = "cuda:0"
sc = make_scene(39000, "fr3_office", seed=11, near_fraction=0.0, device=dev)
intr = Intrinsics(sc.intr, dev)
vp = Viewpoint(0, torch.rand(3, intr.height, intr.width, device=dev), torch.ones(intr.height, intr.width, device=dev), dev)
vp.update_RT(sc.R.to(dev).contiguous(), sc.t.to(dev).contiguous())
P = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.rotations, sc.scales, sc.opacities, sc.colors)]
bg = torch.zeros(3, device=dev)


def step():
    pkg = render(vp, intr, *P, bg)
    loss = fused_losses.get_loss_mapping(pkg["render"], pkg["depth"], vp)
    loss.backward()
    for p in P:
        p.grad = None


for _ in range(20):
    step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"eager exact-count (one count read-back + sync per forward, as upstream): {(time.perf_counter() - t0) / n * 1e3:.3f} ms / iteration")
# where the exact path's wall time goes: host stamps around the three calls of an iteration (the read-back inside render()
# waits for the GPU to drain everything queued so far, so `render` contains the previous iteration's backward on the device)
acc = [0.0, 0.0, 0.0, 0.0]
for _ in range(n):
    a = time.perf_counter()
    pkg = render(vp, intr, *P, bg)
    b = time.perf_counter()
    loss = fused_losses.get_loss_mapping(pkg["render"], pkg["depth"], vp)
    c = time.perf_counter()
    loss.backward()
    d = time.perf_counter()
    for p in P:
        p.grad = None
    e = time.perf_counter()
    for i, v in enumerate((b - a, c - b, d - c, e - d)):
        acc[i] += v
torch.cuda.synchronize()
print("exact-count host stamps per iteration: render %.3f ms (contains the count read-back), loss forward %.3f, loss.backward() %.3f, "
      "grad reset %.3f" % tuple(1e3 * v / n for v in acc))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
torch.cuda.synchronize()
pr.disable()
print("---- cProfile, exact-count path")
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
# capacity mode: nothing synchronises, so the wall time per iteration is max(host time, GPU time); the GPU time of the same
# iteration comes from events around a burst the host has queued ahead
from monogs_amd import rasterizer as R
R.set_sync_free(True)
for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
t_host = (time.perf_counter() - t0) / n          # the loop returns when the LAST launch is queued
torch.cuda.synchronize()
t_wall = (time.perf_counter() - t0) / n
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    step()
e1.record()
torch.cuda.synchronize()
print(f"eager capacity mode: host {t_host * 1e3:.3f} ms / iteration to queue, wall {t_wall * 1e3:.3f} ms, "
      f"device span {e0.elapsed_time(e1) / n:.3f} ms / iteration (render + loss + backward, 39 k Gaussians, 640x480)")
assert not R.check_overflow()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
