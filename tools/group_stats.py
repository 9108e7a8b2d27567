"""How many loop trips would the blend walk take with several survivor streams per wave?  (mgs_debug_blend_stats, words 8..22.)

    python tools/group_stats.py            # C5 (2 M Gaussians, 1080p) and 100 k / VGA

Prints, per way of cutting the 8x8 quadrant into pixel groups, trips / survivors for the two pairing disciplines and the
(group, survivor) rows the backward would flush.  Round 5's gate for the two-stream blend kernels: trips <= 0.75 x survivors.
"""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from monogs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, debug_blend_stats  # noqa: E402
from monogs_amd.synthetic import make_scene, scene_settings  # noqa: E402


def run(P, intr, seed=2):
    dev = torch.device("cuda", 0)
    sc = make_scene(P, intr, seed=seed)
    st = scene_settings(sc, GaussianRasterizationSettings, device=dev)
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
    xyz, rgb, opac, scaling, rot = leaf(sc.means3D), leaf(sc.colors), leaf(sc.opacities), leaf(sc.scales), leaf(sc.rotations)
    means2D = torch.zeros_like(xyz, requires_grad=True)
    color, radii, depth, opacity, n_touched = GaussianRasterizer(st)(
        means3D=xyz, means2D=means2D, opacities=opac, colors_precomp=rgb, scales=scaling, rotations=rot)
    w = debug_blend_stats(color)
    S = max(1, w["survivors"])
    out = {"scene": f"{P} / {intr}", "steps": w["steps"], "survivors": w["survivors"], "active_survivors": w["active_survivors"],
           "active_lanes_per_active_survivor": round(w["active_pairs"] / max(1, w["active_survivors"]), 2)}
    for name, g in w["group_streams"].items():
        out[name] = {"trips_paired_per_step/S": round(g["trips_paired_per_step"] / S, 4),
                     "trips_own_lists/S": round(g["trips_own_lists"] / S, 4), "rows/S": round(g["rows"] / S, 4)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    run(2_000_000, "davis_1080p")
    run(100_000, "fr3_office")
    run(40_000, "fr3_office")
