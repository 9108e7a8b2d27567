#!/usr/bin/env python3
"""How far the HIP gradients are from a float64 evaluation of the same function, next to the float32 oracle's own distance
(BASELINE config 2 with the mapping loss's upstream gradients: tests/test_gpu_parity.py::test_c2_100k_mapping_loss_gradients).
The float64 / float32 oracle results are cached in /tmp so that kernel variants (MGS_LIB_PATH, MGS_DEBUG_OPTIONS) can be
compared in one GPU call:   python tools/grad_accuracy.py [label]"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from monogs_amd import fused_losses
from monogs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from monogs_amd.synthetic import make_scene, scene_settings
from oracle import OracleSettings, rasterize_autograd
from oracle.slam_losses import get_loss_mapping as loss_ref
from test_gpu_parity import _c2_frame, _inputs

DEV = "cuda:0"
label = sys.argv[1] if len(sys.argv) > 1 else "default"
P = int(os.environ.get("ACC_P", "100000"))
sc = make_scene(P, "fr3_office", seed=1)
inp = _inputs(sc)
st = scene_settings(sc, GaussianRasterizationSettings, device=DEV)


def hip(g_color=None, g_depth=None, vp=None):
    leaves = {k: v.to(DEV).clone().requires_grad_(True) for k, v in inp.items()}
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
    th = torch.zeros(3, device=DEV, requires_grad=True)
    rh = torch.zeros(3, device=DEV, requires_grad=True)
    out = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                 colors_precomp=leaves["colors_precomp"], scales=leaves["scales"], rotations=leaves["rotations"],
                                 theta=th, rho=rh)
    if vp is not None:
        fused_losses.get_loss_mapping(out[0], out[2], vp).backward()
    elif g_color is not None:
        torch.autograd.backward([out[0], out[2]], [g_color.to(DEV), g_depth.to(DEV)])
    g = {k: v.grad.cpu() for k, v in leaves.items()} if (vp is not None or g_color is not None) else {}
    if g:
        g.update(means2D=m2.grad.cpu(), theta=th.grad.cpu(), rho=rh.grad.cpu())
    return out, g


cache = f"/tmp/grad_accuracy_{P}.pt"
if os.path.exists(cache):
    c = torch.load(cache)
else:
    out, _ = hip()
    vpc = _c2_frame(out[0].detach().cpu(), out[2].detach().cpu())
    c_ref, d_ref = out[0].detach().cpu().requires_grad_(True), out[2].detach().cpu().requires_grad_(True)
    vpc.exposure_a.requires_grad_(True); vpc.exposure_b.requires_grad_(True)
    gc, gd = torch.autograd.grad(loss_ref(c_ref, d_ref, vpc), [c_ref, d_ref])
    ost = scene_settings(sc, OracleSettings)
    c = dict(rgb=vpc.rgb, depth=vpc.depth, mask=vpc.mask, gc=gc, gd=gd)
    for name, (a, b) in (("loss", (gc, gd)), ("noise", (sc.grad_color, sc.grad_depth))):
        c[name + "64"] = rasterize_autograd(inp, ost, a, b, dtype=torch.float64)[1]
        c[name + "32"] = rasterize_autograd(inp, ost, a, b, dtype=torch.float32)[1]
    torch.save(c, cache)
for name, (a, b) in (("loss", (c["gc"], c["gd"])), ("noise", (sc.grad_color, sc.grad_depth))):
    _, g = hip(a, b)
    row = []
    for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp", "means2D", "theta", "rho"):
        ref = c[name + "64"][k].double()
        e = lambda x: ((x.reshape(ref.shape).double() - ref).norm() / ref.norm()).item()  # noqa: E731
        row.append(f"{k} {e(g[k]):.1e}/{e(c[name + '32'][k]):.1e}")
    print(f"[{label}] upstream = {name:5s} (HIP / float32 oracle, relative L2 vs float64): " + "  ".join(row), flush=True)

# ---- the fused loss's upstream gradients against the PyTorch mirror's, on the same images
out, _ = hip()
color, depth = out[0].detach(), out[2].detach()
vp = types.SimpleNamespace(rgb=c["rgb"].to(DEV), depth=c["depth"].to(DEV), mask=c["mask"].to(DEV),
                           exposure_a=torch.tensor([0.03], device=DEV), exposure_b=torch.tensor([-0.02], device=DEV))
lg = fused_losses.loss_grads(color, depth, None, vp, tracking=False)
torch.cuda.synchronize()
for name, got, ref in (("dL/dcolor", lg.d_render.cpu(), c["gc"]), ("dL/ddepth", lg.d_depth.cpu(), c["gd"])):
    d = (got.double() - ref.double())
    nz = ref != 0
    ratio = (got[nz].double() / ref[nz].double())
    print(f"[{label}] fused loss vs torch mirror, {name}: rel L2 {d.norm() / ref.double().norm():.2e}, elements that differ by more "
          f"than 1e-3 relative: {int((d.abs() > 1e-3 * ref.abs().double().max()).sum())} of {ref.numel()}, "
          f"ratio got / ref on the non-zero ones: min {ratio.min():.7f} max {ratio.max():.7f}; zero pattern equal: {bool(((got != 0) == nz).all())}")
