"""Fused pose step vs torch.optim.Adam + the retraction mirror (itself pinned to the reference's update_pose)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _vp(seed):
    from monogs_amd import camera as cam
    T0 = cam.se3_exp(torch.tensor([0.1, -0.2, 0.3, 0.05, 0.02, -0.04]))
    P = lambda n, v=0.0: torch.nn.Parameter(torch.full((n,), v, device=DEV))  # noqa: E731
    return types.SimpleNamespace(R=T0[:3, :3].contiguous().to(DEV), T=T0[:3, 3].contiguous().to(DEV),
                                 cam_rot_delta=P(3), cam_trans_delta=P(3), exposure_a=P(1, 0.02), exposure_b=P(1, -0.01))


def test_pose_step_matches_adam_plus_update_pose(native_lib):
    from monogs_amd import camera as cam
    from monogs_amd.pose_optim import PoseAdam
    a, b = _vp(0), _vp(0)
    fused = PoseAdam(a)
    ref = torch.optim.Adam([{"params": [b.cam_rot_delta], "lr": 0.003}, {"params": [b.cam_trans_delta], "lr": 0.001},
                            {"params": [b.exposure_a], "lr": 0.01}, {"params": [b.exposure_b], "lr": 0.01}])
    g = torch.Generator().manual_seed(0)
    for it in range(25):
        scale = 10.0 ** (-it / 6.0)
        grads = [torch.randn(n, generator=g).to(DEV) * scale for n in (3, 3, 1, 1)]
        for vp in (a, b):
            for p, gr in zip((vp.cam_rot_delta, vp.cam_trans_delta, vp.exposure_a, vp.exposure_b), grads):
                p.grad = gr.clone()
        conv_f = fused.step_and_retract()
        ref.step()
        with torch.no_grad():
            Rn, Tn, conv_r = cam.retract_pose(b.R, b.T, b.cam_trans_delta.data, b.cam_rot_delta.data)
            b.R, b.T = Rn, Tn
            b.cam_rot_delta.data.zero_()
            b.cam_trans_delta.data.zero_()
        assert conv_f == conv_r
        assert torch.allclose(a.R, b.R, atol=2e-6) and torch.allclose(a.T, b.T, atol=2e-6)
        assert torch.allclose(a.exposure_a, b.exposure_a, atol=1e-6) and torch.allclose(a.exposure_b, b.exposure_b, atol=1e-6)
        assert a.cam_rot_delta.abs().max() == 0 and a.cam_trans_delta.abs().max() == 0
    # rotation stays orthonormal
    assert torch.allclose(a.R @ a.R.t(), torch.eye(3, device=DEV), atol=1e-5)


def test_pose_step_sticky_convergence(native_lib):
    """MGS_POSE_STICKY: after the first converged update every further call is a no-op until reset()."""
    from monogs_amd.pose_optim import PoseAdam
    a = _vp(1)
    opt = PoseAdam(a, sticky=True)
    big, tiny = torch.full((3,), 0.5, device=DEV), torch.full((3,), 1e-12, device=DEV)

    def step(g, thr):
        a.cam_rot_delta.grad, a.cam_trans_delta.grad = g.clone(), g.clone()
        a.exposure_a.grad, a.exposure_b.grad = g[:1].clone(), g[:1].clone()
        return opt.step_and_retract(converged_threshold=thr)

    assert step(big, 1e-4) is False
    assert step(big, 1.0) is True                      # |tau| < 1: converged
    snap = [t.clone() for t in (a.R, a.T, a.exposure_a.data, a.exposure_b.data, opt.m, opt.v, opt.t_dev)]
    assert step(big, 1e-4) is True                     # sticky: nothing moves, flag stays
    for t, s in zip((a.R, a.T, a.exposure_a.data, a.exposure_b.data, opt.m, opt.v, opt.t_dev), snap):
        assert torch.equal(t, s)
    opt.reset()
    assert int(opt.t_dev) == 0 and step(big, 1e-4) is False and int(opt.t_dev) == 1
    assert not torch.equal(a.R, snap[0])
