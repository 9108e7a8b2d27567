"""How much of the blend kernels' time is ramp-up / tail?  The same scene density at 1080p (8 160 tiles on 2 048 workgroup slots =
4 rounds) and at 2160p (4 x the Gaussians, 4 x the tiles = 16 rounds): per-survivor time that falls with the round count is tail.

    python tools/tail_probe.py [steps]
"""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monogs_amd.camera import INTRINSICS  # noqa: E402
from monogs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, collect_timing, debug_blend_stats  # noqa: E402
from monogs_amd.synthetic import make_scene, scene_settings  # noqa: E402


def run(P, intr, steps):
    dev = torch.device("cuda", 0)
    sc = make_scene(P, intr, seed=2)
    st = scene_settings(sc, GaussianRasterizationSettings, device=dev)
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
    xyz, rgb, opac, scaling, rot = leaf(sc.means3D), leaf(sc.colors), leaf(sc.opacities), leaf(sc.scales), leaf(sc.rotations)
    gc, gd = sc.grad_color.to(dev), sc.grad_depth.to(dev)
    r = GaussianRasterizer(st)
    surv = None
    f, b = [], []
    for i in range(steps + 3):
        for p in (xyz, rgb, opac, scaling, rot):
            p.grad = None
        m2 = torch.zeros_like(xyz, requires_grad=True)
        with collect_timing() as sink:
            out = r(means3D=xyz, means2D=m2, opacities=opac, colors_precomp=rgb, scales=scaling, rotations=rot)
            if surv is None:
                surv = debug_blend_stats(out[0])["survivors"]
            torch.autograd.backward([out[0], out[2]], [gc, gd])
        if i >= 3:
            f.append(sum(d.get("blend_fwd_ms", 0.0) for d in sink))
            b.append(sum(d.get("blend_bwd_ms", 0.0) for d in sink))
    fm, bm = sorted(f)[len(f) // 2], sorted(b)[len(b) // 2]
    W, H = st.image_width, st.image_height
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    print(json.dumps({"scene": f"{P} / {W}x{H}", "tiles": tiles, "rounds_of_2048_workgroups": round(tiles / 2048, 2), "survivors": surv,
                      "blend_fwd_ms": round(fm, 4), "blend_bwd_ms": round(bm, 4),
                      "fwd_ns_per_survivor": round(fm * 1e6 * 1024 / surv, 2), "bwd_ns_per_survivor": round(bm * 1e6 * 1024 / surv, 2)}), flush=True)


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    INTRINSICS["synthetic_2160p"] = dict(fx=1920.0, fy=1920.0, cx=1920.0, cy=1080.0, W=3840, H=2160)
    INTRINSICS["synthetic_540p"] = dict(fx=480.0, fy=480.0, cx=480.0, cy=270.0, W=960, H=540)
    run(500_000, "synthetic_540p", steps)
    run(2_000_000, "davis_1080p", steps)
    run(8_000_000, "synthetic_2160p", steps)
