#!/usr/bin/env python3
"""Which stage of a C5 step is slower in the first steps of a process?  Per-stage device ms (HIP events between the stages) of
the first N forward+backward steps after start-up, exact path.  (bench.py's 20-step runs read ~4 % below its 6 000-step runs;
neither a 400-ms preload of the ALUs and the memory system nor a bounded run-ahead of the host changes that.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from monogs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, collect_timing
from monogs_amd.synthetic import make_scene, scene_settings

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sc = make_scene(2_000_000, "davis_1080p", seed=2)
st = scene_settings(sc, GaussianRasterizationSettings, device=dev)
leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
xyz, rgb, opac, scaling, rot = leaf(sc.means3D), leaf(sc.colors), leaf(sc.opacities), leaf(sc.scales), leaf(sc.rotations)
theta, rho = torch.zeros(3, device=dev, requires_grad=True), torch.zeros(3, device=dev, requires_grad=True)
gc, gd = sc.grad_color.to(dev), sc.grad_depth.to(dev)
rows = []
with collect_timing() as sink:
    for i in range(n):
        for p in (xyz, rgb, opac, scaling, rot, theta, rho):
            p.grad = None
        m2 = torch.zeros_like(xyz, requires_grad=True)
        out = GaussianRasterizer(st)(means3D=xyz, means2D=m2, opacities=opac, colors_precomp=rgb, scales=scaling, rotations=rot,
                                     theta=theta, rho=rho)
        torch.autograd.backward([out[0], out[2]], [gc, gd])
        torch.cuda.synchronize()
keys = ("preprocess_ms", "depth_sort_ms", "duplicate_ms", "sort_ms", "blend_fwd_ms", "blend_bwd_ms", "geom_bwd_ms")
print("step  " + "  ".join(f"{k[:-3]:>11s}" for k in keys) + "        sum")
for i in range(n):
    d = dict(sink[2 * i])
    d.update({k: v for k, v in sink[2 * i + 1].items() if isinstance(v, float) and v > 0})
    print(f"{i:4d}  " + "  ".join(f"{d.get(k, 0.0):11.4f}" for k in keys) + f"  {sum(d.get(k, 0.0) for k in keys):9.4f}")
