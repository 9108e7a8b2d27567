"""Fused Adam over the Gaussian parameter groups and the per-keyframe densification statistics.

``GaussianAdam`` is step-for-step ``torch.optim.Adam(groups, lr=0.0, eps=1e-15)`` as the reference builds it
(/root/reference/gaussian_splatting/scene/gaussian_model.py:398-442) in one launch with the step count on the
device (graph-capturable); ``add_densification_stats`` is gaussian_model.py:888-892 plus the ``max_radii_2d``
update of /root/reference/utils/slam_mapper.py:453-457.  Checked against torch in tests/test_gpu_optim.py.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib
from .rasterizer import _stream


class GaussianAdam:
    def __init__(self, params: Sequence[torch.Tensor], lrs: Sequence[float], betas=(0.9, 0.999), eps=1e-15):
        assert 1 <= len(params) <= 8 and len(params) == len(lrs)
        self.params = list(params)
        self.lrs = [float(x) for x in lrs]
        self.betas, self.eps = betas, float(eps)
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=self.params[0].device)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        lib = _lib.load()
        n = len(self.params)
        vp = C.c_void_p * n
        grads = [None if p.grad is None else p.grad.contiguous() for p in self.params]
        tab = lambda ts: vp(*[None if t is None else t.data_ptr() for t in ts])  # noqa: E731
        numel = (C.c_uint64 * n)(*[p.numel() for p in self.params])
        lr = (C.c_float * n)(*self.lrs)
        with torch.cuda.device(self.params[0].device):
            _lib.check(lib.mgs_adam_step(n, tab(self.params), tab(grads), tab(self.exp_avg), tab(self.exp_avg_sq), numel,
                                         lr, self.betas[0], self.betas[1], self.eps, 0, self.t_dev.data_ptr(), _stream()),
                       "mgs_adam_step")


@torch.no_grad()
def add_densification_stats(viewspace_grad: torch.Tensor, radii: torch.Tensor, xyz_gradient_accum=None, denom=None,
                            max_radii_2d=None):
    """In-place update of the three statistics for one rendered keyframe (visible = radii > 0)."""
    lib = _lib.load()
    P = radii.shape[0]
    g = viewspace_grad.contiguous()
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with torch.cuda.device(radii.device):
        _lib.check(lib.mgs_densify_stats(P, g.data_ptr(), radii.contiguous().data_ptr(), p(xyz_gradient_accum), p(denom),
                                         p(max_radii_2d), _stream()), "mgs_densify_stats")
