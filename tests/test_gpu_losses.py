"""Fused HIP losses vs the PyTorch mirrors (which are pinned to the reference's own outputs in
tests/test_golden.py) and vs the committed golden vectors themselves (pytest -m gpu)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = os.path.join(os.path.dirname(__file__), "golden")


def _vp(H, W, seed, device):
    g = torch.Generator().manual_seed(seed)
    depth = torch.rand(H, W, generator=g) * 3
    depth[torch.rand(H, W, generator=g) < 0.2] = 0
    vp = types.SimpleNamespace(
        rgb=torch.rand(3, H, W, generator=g).to(device), depth=depth.to(device),
        mask=(torch.rand(H, W, generator=g) > 0.1).to(device), grad_mask=(torch.rand(H, W, generator=g) > 0.4).to(device),
        exposure_a=torch.tensor([0.07], device=device, requires_grad=True),
        exposure_b=torch.tensor([-0.03], device=device, requires_grad=True))
    render = torch.rand(3, H, W, generator=g).to(device).requires_grad_(True)
    rdepth = (torch.rand(1, H, W, generator=g) * 3).to(device).requires_grad_(True)
    op = torch.rand(1, H, W, generator=g)
    op[torch.rand(1, H, W, generator=g) < 0.6] = 0.995
    return vp, render, rdepth, op.to(device)


def _grads(loss, xs):
    return torch.autograd.grad(loss, xs, allow_unused=True)


@pytest.mark.parametrize("H,W", [(24, 32), (480, 640), (237, 325)])
@pytest.mark.parametrize("init", [False, True])
def test_fused_mapping_loss(native_lib, H, W, init):
    from monogs_amd import fused_losses as F
    from oracle import slam_losses as S
    vp, render, rdepth, _ = _vp(H, W, 1, DEV)
    xs = [render, rdepth] + ([] if init else [vp.exposure_a, vp.exposure_b])
    lf = F.get_loss_mapping(render, rdepth, vp, init=init)
    lr = S.get_loss_mapping(render, rdepth, vp, init=init)
    assert abs(lf.item() - lr.item()) <= 2e-6 * max(1.0, abs(lr.item()))
    for a, b in zip(_grads(lf * 1.7, xs), _grads(lr * 1.7, xs)):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-9), (a - b).abs().max()


@pytest.mark.parametrize("H,W", [(24, 32), (480, 640)])
def test_fused_tracking_loss(native_lib, H, W):
    from monogs_amd import fused_losses as F
    from oracle import slam_losses as S
    vp, render, rdepth, op = _vp(H, W, 2, DEV)
    xs = [render, rdepth, vp.exposure_a, vp.exposure_b]
    lf = F.get_loss_tracking(render, rdepth, op, vp)
    lr = S.get_loss_tracking(render, rdepth, op, vp)
    assert abs(lf.item() - lr.item()) <= 2e-6 * max(1.0, abs(lr.item()))
    for a, b in zip(_grads(lf, xs), _grads(lr, xs)):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-9), (a - b).abs().max()
    # empty depth mask: the depth term is exactly zero
    op0 = torch.zeros_like(op)
    assert F.get_loss_tracking(render, rdepth, op0, vp).item() == 0.0


def test_fused_losses_against_reference_golden(native_lib):
    from monogs_amd import fused_losses as F
    d = np.load(os.path.join(G, "losses.npz"))
    t = lambda k: torch.tensor(d[k]).to(DEV)  # noqa: E731
    vp = types.SimpleNamespace(rgb=t("gt_rgb"), depth=t("gt_depth"), mask=t("gt_mask"), grad_mask=t("grad_mask"),
                               exposure_a=torch.tensor([float(d["exposure"][0])], device=DEV),
                               exposure_b=torch.tensor([float(d["exposure"][1])], device=DEV))
    render, depth = t("render").requires_grad_(True), t("depth").requires_grad_(True)
    for tag, init in (("map", False), ("init", True)):
        loss = F.get_loss_mapping(render, depth, vp, init=init)
        gr, gd = torch.autograd.grad(loss, [render, depth])
        assert abs(loss.item() - d[f"loss_{tag}"][0]) < 2e-6
        assert np.allclose(gr.cpu().numpy(), d[f"grad_render_{tag}"], atol=1e-8)
        assert np.allclose(gd.cpu().numpy(), d[f"grad_depth_{tag}"], atol=1e-8)
    loss = F.get_loss_tracking(render, depth, t("opacity"), vp)
    gr, gd = torch.autograd.grad(loss, [render, depth])
    assert abs(loss.item() - d["loss_track"][0]) < 2e-6
    assert np.allclose(gr.cpu().numpy(), d["grad_render_track"], atol=1e-8)
    assert np.allclose(gd.cpu().numpy(), d["grad_depth_track"], atol=1e-8)


def test_fused_losses_invert_depth_against_reference_golden(native_lib):
    """`invert_depth=True` (/root/reference/utils/slam_utils.py:83-88, :138-141) against the reference's own outputs."""
    from monogs_amd import fused_losses as F
    d = np.load(os.path.join(G, "losses_invert.npz"))
    t = lambda k: torch.tensor(d[k]).to(DEV)  # noqa: E731
    vp = types.SimpleNamespace(rgb=t("gt_rgb"), depth=t("gt_depth"), mask=t("gt_mask"), grad_mask=t("grad_mask"),
                               exposure_a=torch.tensor([float(d["exposure"][0])], device=DEV),
                               exposure_b=torch.tensor([float(d["exposure"][1])], device=DEV))
    render, depth = t("render").requires_grad_(True), t("depth").requires_grad_(True)
    loss = F.get_loss_mapping(render, depth, vp, invert_depth=True)
    gr, gd = torch.autograd.grad(loss, [render, depth])
    assert abs(loss.item() - d["loss_map"][0]) < 2e-6
    assert np.allclose(gr.cpu().numpy(), d["grad_render_map"], atol=1e-8)
    assert np.allclose(gd.cpu().numpy(), d["grad_depth_map"], rtol=1e-5, atol=1e-8)
    loss = F.get_loss_tracking(render, depth, t("opacity"), vp, invert_depth=True)
    gr, gd = torch.autograd.grad(loss, [render, depth])
    assert abs(loss.item() - d["loss_track"][0]) < 2e-6
    assert np.allclose(gr.cpu().numpy(), d["grad_render_track"], atol=1e-8)
    assert np.allclose(gd.cpu().numpy(), d["grad_depth_track"], rtol=1e-5, atol=1e-8)
    lg = F.loss_grads(render, depth, None, vp, tracking=False, invert_depth=True)       # the two-launch path
    assert np.allclose(lg.d_depth.cpu().numpy(), d["grad_depth_map"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("tracking,init", [(True, False), (False, False), (False, True)])
@pytest.mark.parametrize("H,W", [(480, 640), (237, 325)])
def test_loss_grads_two_launch_path_equals_the_autograd_loss(native_lib, tracking, init, H, W):
    """`loss_grads` (value + upstream gradients in two launches, no autograd node for the scalar -- what the captured
    tracking / mapping iterations use) gives the numbers of the autograd loss forward + backward."""
    from monogs_amd import fused_losses as F
    vp, render, rdepth, op = _vp(H, W, 31, DEV)
    if tracking:
        loss = F.get_loss_tracking(render, rdepth, op, vp)
    else:
        loss = F.get_loss_mapping(render, rdepth, vp, init=init)
    xs = [render, rdepth] + ([] if init else [vp.exposure_a, vp.exposure_b])
    ref = _grads(loss, xs)
    lg = F.loss_grads(render, rdepth, op if tracking else None, vp, tracking=tracking, init=init)
    assert torch.allclose(lg.loss, loss.detach(), rtol=1e-6, atol=0)
    assert torch.equal(lg.d_render, ref[0]) and torch.equal(lg.d_depth, ref[1])
    if init:
        assert lg.d_exposure_a is None
    else:           # (scale x the forward's unscaled sums, stored by one thread: rounding differs from autograd's order)
        assert torch.allclose(lg.d_exposure_a, ref[2], rtol=1e-5, atol=1e-9)
        assert torch.allclose(lg.d_exposure_b, ref[3], rtol=1e-5, atol=1e-9)
