#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING the reference's own helpers
(the ones that import on CPU here).  Run in the build container only:

    python tests/golden/make_golden.py          # needs /root/reference

Outputs (data only -- inputs and the reference's outputs):
  camera_pose.npz : getProjectionMatrix / getWorld2View / CameraIntrinsics.FoV / SE3_exp / update_pose
  sh_eval.npz     : eval_sh for degrees 0..3 on seeded inputs
  losses.npz      : get_loss_mapping / get_loss_tracking values and autograd gradients on seeded images
  losses_invert.npz: the same with invert_depth=True
  lr_schedule.npz : general_utils.helper (the xyz learning-rate schedule of update_learning_rate)
  median_depth.npz: get_median_depth (value, std, valid mask) on a seeded depth image with holes, with and without a mask
  grad_mask.npz   : image_gradient / image_gradient_mask combined as compute_grad_mask does (the tracking loss's edge mask)
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

from gaussian_splatting.utils.graphics_utils import getProjectionMatrix, getWorld2View  # noqa: E402
from gaussian_splatting.utils.sh_utils import eval_sh  # noqa: E402
from utils.pose_utils import SE3_exp, SO3_exp, V, update_pose  # noqa: E402
from utils.slam_utils import get_loss_mapping, get_loss_tracking, get_median_depth  # noqa: E402

INTR = {
    "fr3_office": dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6, W=640, H=480),
    "replica": dict(fx=600.0, fy=600.0, cx=599.5, cy=339.5, W=1200, H=680),
    "davis_1080p": dict(fx=960.0, fy=960.0, cx=960.0, cy=540.0, W=1920, H=1080),
}
TAUS = np.array([
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
    [0.1, -0.2, 0.3, 0.05, 0.02, -0.04],
    [-0.5, 0.25, 1.5, 0.7, -0.3, 0.2],
    [1e-3, 2e-3, -1e-3, 1e-6, -2e-6, 3e-6],      # small-angle branch (|theta| < 1e-5)
    [0.3, 0.1, -0.2, 1.2, 0.9, -1.4],
], dtype=np.float32)


def camera_pose():
    out = {"taus": TAUS}
    se3 = []
    for tau in TAUS:
        se3.append(SE3_exp(torch.tensor(tau)).numpy())
    out["se3_exp"] = np.stack(se3)
    out["so3_exp"] = np.stack([SO3_exp(torch.tensor(t[3:])).numpy() for t in TAUS])
    out["V"] = np.stack([V(torch.tensor(t[3:])).numpy() for t in TAUS])
    for name, k in INTR.items():
        fx = torch.nn.Parameter(torch.tensor([k["fx"]]), requires_grad=False)
        fy = torch.nn.Parameter(torch.tensor([k["fy"]]), requires_grad=False)
        P = getProjectionMatrix(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=k["cx"], cy=k["cy"], W=k["W"], H=k["H"])
        out[f"{name}_projection_T"] = P.transpose(0, 1).detach().numpy()
        # /root/reference/utils/camera_utils.py:30-36
        fovx = 2 * torch.atan(k["W"] / (2 * fx)).cpu().item()
        fovy = 2 * torch.atan(k["H"] / (2 * fy)).cpu().item()
        out[f"{name}_tanfov"] = np.array([math.tan(fovx * 0.5), math.tan(fovy * 0.5)], dtype=np.float64)
        views, fulls, centers = [], [], []
        for tau in TAUS:
            T = SE3_exp(torch.tensor(tau))
            w2c_T = getWorld2View(T[:3, :3], T[:3, 3]).transpose(0, 1)
            full = (w2c_T.unsqueeze(0).bmm(P.transpose(0, 1).detach().unsqueeze(0))).squeeze(0)
            views.append(w2c_T.numpy())
            fulls.append(full.numpy())
            centers.append(w2c_T.inverse()[3, :3].numpy())
        out[f"{name}_view_T"] = np.stack(views)
        out[f"{name}_full_T"] = np.stack(fulls)
        out[f"{name}_campos"] = np.stack(centers)
    # update_pose: left-multiplicative retraction and the convergence flag
    cam = types.SimpleNamespace()
    T0 = SE3_exp(torch.tensor(TAUS[1]))
    cam.R, cam.T = T0[:3, :3].clone(), T0[:3, 3].clone()
    cam.cam_trans_delta = torch.nn.Parameter(torch.tensor([0.01, -0.02, 0.005]))
    cam.cam_rot_delta = torch.nn.Parameter(torch.tensor([0.002, 0.001, -0.003]))
    cam.update_RT = lambda R, t: (setattr(cam, "R", R), setattr(cam, "T", t))
    out["update_in_R"], out["update_in_T"] = T0[:3, :3].numpy(), T0[:3, 3].numpy()
    out["update_rho"] = cam.cam_trans_delta.detach().numpy().copy()
    out["update_theta"] = cam.cam_rot_delta.detach().numpy().copy()
    conv = update_pose(cam)
    out["update_out_R"], out["update_out_T"] = cam.R.detach().numpy(), cam.T.detach().numpy()
    out["update_converged"] = np.array([bool(conv)])
    np.savez_compressed(os.path.join(HERE, "camera_pose.npz"), **out)


def sh_eval():
    g = torch.Generator().manual_seed(123)
    n = 64
    dirs = torch.randn(n, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    sh = torch.randn(n, 3, 16, generator=g)          # [..., C, coeffs] as eval_sh expects
    out = {"dirs": dirs.numpy(), "sh": sh.numpy()}
    for deg in range(4):
        out[f"deg{deg}"] = eval_sh(deg, sh, dirs).numpy()
    np.savez_compressed(os.path.join(HERE, "sh_eval.npz"), **out)


def losses():
    g = torch.Generator().manual_seed(7)
    H, W = 24, 32
    vp = types.SimpleNamespace()
    vp.rgb = torch.rand(3, H, W, generator=g)
    vp.depth = torch.rand(H, W, generator=g) * 3.0
    vp.depth[torch.rand(H, W, generator=g) < 0.2] = 0.0             # invalid depth pixels
    vp.mask = torch.rand(H, W, generator=g) > 0.1
    vp.exposure_a = torch.tensor([0.05])
    vp.exposure_b = torch.tensor([-0.02])
    render = torch.rand(3, H, W, generator=g).requires_grad_(True)
    depth = (torch.rand(1, H, W, generator=g) * 3.0).requires_grad_(True)
    out = {"gt_rgb": vp.rgb.numpy(), "gt_depth": vp.depth.numpy(), "gt_mask": vp.mask.numpy(),
           "exposure": np.array([0.05, -0.02], dtype=np.float32),
           "render": render.detach().numpy(), "depth": depth.detach().numpy()}
    for init in (False, True):
        loss = get_loss_mapping(render, depth, vp, init=init)
        gr, gd = torch.autograd.grad(loss, [render, depth])
        tag = "init" if init else "map"
        out[f"loss_{tag}"] = np.array([loss.item()], dtype=np.float64)
        out[f"grad_render_{tag}"] = gr.numpy()
        out[f"grad_depth_{tag}"] = gd.numpy()
    # tracking loss (needs grad_mask and the rendered opacity; > 0.99 on part of the image)
    vp.grad_mask = torch.rand(H, W, generator=g) > 0.4
    opacity = torch.rand(1, H, W, generator=g)
    opacity[torch.rand(1, H, W, generator=g) < 0.6] = 0.995
    opacity = opacity.requires_grad_(True)
    out["grad_mask"], out["opacity"] = vp.grad_mask.numpy(), opacity.detach().numpy()
    loss = get_loss_tracking(render, depth, opacity, vp)
    gr, gd, go = torch.autograd.grad(loss, [render, depth, opacity])
    out["loss_track"] = np.array([loss.item()], dtype=np.float64)
    out["grad_render_track"], out["grad_depth_track"], out["grad_opacity_track"] = gr.numpy(), gd.numpy(), go.numpy()
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)


def losses_invert():
    """The `invert_depth=True` branches (slam_utils.py:83-88 with eps = 1e-6, :138-141 without)."""
    g = torch.Generator().manual_seed(17)
    H, W = 24, 32
    vp = types.SimpleNamespace()
    vp.rgb = torch.rand(3, H, W, generator=g)
    vp.depth = torch.rand(H, W, generator=g) * 3.0 + 0.3
    vp.depth[torch.rand(H, W, generator=g) < 0.2] = 0.0
    vp.mask = torch.rand(H, W, generator=g) > 0.1
    vp.grad_mask = torch.rand(H, W, generator=g) > 0.4
    vp.exposure_a = torch.tensor([0.05])
    vp.exposure_b = torch.tensor([-0.02])
    render = torch.rand(3, H, W, generator=g).requires_grad_(True)
    depth = (torch.rand(1, H, W, generator=g) * 3.0 + 0.3).requires_grad_(True)
    opacity = torch.rand(1, H, W, generator=g)
    opacity[torch.rand(1, H, W, generator=g) < 0.6] = 0.995
    out = {"gt_rgb": vp.rgb.numpy(), "gt_depth": vp.depth.numpy(), "gt_mask": vp.mask.numpy(),
           "grad_mask": vp.grad_mask.numpy(), "opacity": opacity.numpy(),
           "exposure": np.array([0.05, -0.02], dtype=np.float32),
           "render": render.detach().numpy(), "depth": depth.detach().numpy()}
    loss = get_loss_mapping(render, depth, vp, init=False, invert_depth=True)
    gr, gd = torch.autograd.grad(loss, [render, depth])
    out["loss_map"], out["grad_render_map"], out["grad_depth_map"] = np.array([loss.item()]), gr.numpy(), gd.numpy()
    loss = get_loss_tracking(render, depth, opacity, vp, invert_depth=True)
    gr, gd = torch.autograd.grad(loss, [render, depth])
    out["loss_track"], out["grad_render_track"], out["grad_depth_track"] = np.array([loss.item()]), gr.numpy(), gd.numpy()
    np.savez_compressed(os.path.join(HERE, "losses_invert.npz"), **out)


def median_depth():
    g = torch.Generator().manual_seed(77)
    depth = torch.rand(1, 48, 64, generator=g) * 5.0
    depth[torch.rand(1, 48, 64, generator=g) < 0.2] = 0.0           # invalid pixels
    mask = torch.rand(1, 48, 64, generator=g) < 0.6
    out = {"depth": depth.numpy(), "mask": mask.numpy()}
    for tag, m in (("nomask", None), ("mask", mask)):
        med, std, valid = get_median_depth(depth, m, return_std=True)
        out[f"median_{tag}"] = np.array([med.item()], dtype=np.float32)
        out[f"std_{tag}"] = np.array([std.item()], dtype=np.float32)
        out[f"valid_{tag}"] = valid.numpy()
        assert float(get_median_depth(depth, m)) == med.item()
    np.savez_compressed(os.path.join(HERE, "median_depth.npz"), **out)


def lr_schedule():
    from gaussian_splatting.utils.general_utils import helper
    steps = np.array([-1, 0, 1, 10, 100, 1234, 30000, 999999, 1000000, 2000000], dtype=np.int64)
    cases = [  # (lr_init, lr_final, lr_delay_steps, lr_delay_mult, max_steps): MonoGS's xyz group and two delay variants
        (1.6e-4 * 6.0, 1.6e-6 * 6.0, 0, 0.01, 30000),
        (1.6e-4, 1.6e-6, 500, 0.01, 30000),
        (0.0, 0.0, 0, 1.0, 1000),
    ]
    out = {"steps": steps, "cases": np.array(cases, dtype=np.float64)}
    for i, (a, b, ds, dm, ms) in enumerate(cases):
        out[f"lr_{i}"] = np.array([float(helper(int(s_), a, b, int(ds), dm, int(ms))) for s_ in steps], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "lr_schedule.npz"), **out)


def grad_mask():
    """The tracking loss's edge mask: ``image_gradient`` / ``image_gradient_mask`` (/root/reference/utils/slam_utils.py:6-40)
    run as they are, combined as ``CameraExtrinsics.compute_grad_mask`` combines them (utils/camera_utils.py:185-216; that
    method needs a dataset object, its eight lines of tensor arithmetic are repeated here around the reference's two functions).
    Both functions create their filter taps with ``device="cuda"``; there is no GPU in this container, so for the duration of
    the calls the module's ``torch.tensor`` / ``torch.ones`` are wrapped to drop that one keyword -- placement only, the
    convolutions, paddings and comparisons that produce the values are the reference's own."""
    import utils.slam_utils as su

    class _CpuTorch:
        def __getattr__(self, name):
            f = getattr(torch, name)
            if name in ("tensor", "ones"):
                return lambda *a, **k: f(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
            return f
    g = torch.Generator().manual_seed(404)
    imgs = []
    # a smooth image with edges, one with a black border / black patches (the 3x3 validity mask matters), plain noise
    yy, xx = torch.meshgrid(torch.arange(60.0), torch.arange(80.0), indexing="ij")
    a = torch.stack([0.5 + 0.4 * torch.sin(xx / 7.0) * torch.cos(yy / 5.0), 0.5 + 0.3 * torch.tanh((xx - 40) / 2.0) * 0.9,
                     0.3 + 0.6 * ((xx // 10 + yy // 10) % 2)]).clamp(0, 1)
    b = a.clone(); b[:, :6] = 0; b[:, :, -9:] = 0; b[:, 30:40, 20:33] = 0
    c = torch.rand(3, 60, 80, generator=g)
    imgs = [a, b, c]
    out = {"images": torch.stack(imgs).numpy()}
    real = su.torch
    su.torch = _CpuTorch()
    try:
        for i, rgb in enumerate(imgs):
            gray = rgb.mean(dim=0, keepdim=True)
            gv, gh = su.image_gradient(gray)
            mv, mh = su.image_gradient_mask(gray)
            inten = torch.sqrt((gv * mv) ** 2 + (gh * mh) ** 2)
            out[f"intensity_{i}"] = inten.numpy()
            out[f"mask_{i}"] = (inten > inten.median() * 1.1).numpy()        # edge_threshold = 1.1 (camera_utils.py:186)
    finally:
        su.torch = real
    np.savez_compressed(os.path.join(HERE, "grad_mask.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("median_depth", "lr_schedule", "losses_invert", "grad_mask"):   # add one fixture without rewriting the others
        {"median_depth": median_depth, "lr_schedule": lr_schedule, "losses_invert": losses_invert, "grad_mask": grad_mask}[sys.argv[1]]()
        sys.exit(0)
    camera_pose()
    sh_eval()
    losses()
    losses_invert()
    median_depth()
    lr_schedule()
    grad_mask()
    print("golden fixtures written to", HERE)
