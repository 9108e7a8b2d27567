"""Fused ``get_loss_mapping`` / ``get_loss_tracking`` (HIP, forward value + analytic gradients).

Same signatures and values as /root/reference/utils/slam_utils.py:58-146 (mirrored in plain PyTorch in
``oracle/slam_losses.py`` -- test infrastructure -- and checked there against the reference's own outputs); one reduction kernel
per forward and one elementwise kernel per backward instead of ~60 small kernels and two host syncs.
One difference, invisible to the rasteriser: the opacity image receives no gradient from the tracking loss (the
rasteriser ignores dL/dopacity anyway).  ``invert_depth`` (slam_utils.py:83-88, :138-141) is the mode bit
``MGS_LOSS_INVERT_DEPTH`` of the kernels.
"""
from __future__ import annotations

import torch

from . import _lib
from .rasterizer import _f32, _stream, _device_guard


_TRACKING, _INVERT = 1, 2     # MGS_LOSS_TRACKING, MGS_LOSS_INVERT_DEPTH (include/monogs_raster.h)
_DAB = 10     # MGS_LOSS_SCRATCH_DAB
_LOSS = 12    # MGS_LOSS_SCRATCH_LOSS


def _u8(t):
    if t is None:
        return None
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)
    return (t != 0).to(torch.uint8).contiguous()


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, render, depth, opacity, exp_a, exp_b, gt_rgb, gt_depth, mask, grad_mask, mode, init, lam):
        lib = _lib.load()
        render = _f32(render.detach(), "render_image")
        depth = _f32(depth.detach(), "render_depth")
        H, W = render.shape[-2:]
        dev = render.device
        opac = _f32(opacity.detach(), "render_opacity") if opacity is not None else None
        gt_rgb, gt_depth = _f32(gt_rgb, "viewpoint.rgb"), _f32(gt_depth, "viewpoint.depth")
        a = _f32(exp_a.detach(), "exposure_a") if exp_a is not None else None
        b = _f32(exp_b.detach(), "exposure_b") if exp_b is not None else None
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with _device_guard(dev):
            scratch = torch.empty(lib.mgs_loss_scratch_bytes() // 4, dtype=torch.float32, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            _lib.check(lib.mgs_loss_forward(W, H, int(mode), int(init), float(lam), p(render), p(depth), p(opac),
                                            p(gt_rgb), p(gt_depth), p(mask), p(grad_mask), p(a), p(b),
                                            p(scratch), p(loss), _stream()), "mgs_loss_forward")
        ctx.cfg = (W, H, int(mode), int(init), float(lam))
        ctx.has_mask, ctx.has_gm, ctx.has_op, ctx.has_ab = mask is not None, grad_mask is not None, opac is not None, a is not None
        dummy = torch.empty(0, device=dev)
        ctx.save_for_backward(render, depth, opac if opac is not None else dummy, gt_rgb, gt_depth,
                              mask if mask is not None else dummy, grad_mask if grad_mask is not None else dummy,
                              a if a is not None else dummy, b if b is not None else dummy, scratch)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        render, depth, opac, gt_rgb, gt_depth, mask, gm, a, b, scratch = ctx.saved_tensors
        W, H, mode, init, lam = ctx.cfg
        dev = render.device
        p = lambda t, ok=True: t.data_ptr() if ok else None  # noqa: E731
        with _device_guard(dev):
            go = _f32(grad_out.reshape(1), "grad_output")
            d_render = torch.empty_like(render)
            d_depth = torch.empty_like(depth)
            want_ab = ctx.has_ab and not init and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
            d_ab = None
            if want_ab:
                if getattr(ctx, "dab_used", False):        # a second backward of the same forward: separate, cleared buffer
                    d_ab = torch.empty(2, dtype=torch.float32, device=dev)
                else:                                      # the slot the forward left zeroed (MGS_LOSS_SCRATCH_DAB)
                    d_ab = scratch[_DAB:_DAB + 2]
                    ctx.dab_used = True
            _lib.check(lib.mgs_loss_backward(W, H, mode, init, lam, p(render), p(depth), p(opac, ctx.has_op),
                                             p(gt_rgb), p(gt_depth), p(mask, ctx.has_mask), p(gm, ctx.has_gm),
                                             p(a, ctx.has_ab), p(b, ctx.has_ab), p(scratch), p(go), p(d_render),
                                             p(d_depth), d_ab.data_ptr() if d_ab is not None else None, _stream()),
                       "mgs_loss_backward")
        d_a = d_ab[0:1] if d_ab is not None else None      # views: autograd takes them as .grad without a copy kernel
        d_b = d_ab[1:2] if d_ab is not None else None
        return (d_render, d_depth, None, d_a, d_b, None, None, None, None, None, None, None)


def get_loss_mapping(render_image, render_depth, viewpoint, init=False, invert_depth=False, lambda_depth=0.9):
    return _FusedLoss.apply(render_image, render_depth, None, viewpoint.exposure_a, viewpoint.exposure_b,
                            viewpoint.rgb, viewpoint.depth, _u8(viewpoint.mask), None,
                            _INVERT if invert_depth else 0, bool(init), float(lambda_depth))


def get_loss_tracking(render_image, render_depth, render_opacity, viewpoint, invert_depth=False, lambda_depth=0.9):
    return _FusedLoss.apply(render_image, render_depth, render_opacity, viewpoint.exposure_a, viewpoint.exposure_b,
                            viewpoint.rgb, viewpoint.depth, _u8(viewpoint.mask), _u8(viewpoint.grad_mask),
                            _TRACKING | (_INVERT if invert_depth else 0), False, 0.9)


class LossGrads:
    """Result of ``loss_grads``: upstream gradients for the rasteriser + views of the scalar results on the device."""
    __slots__ = ("d_render", "d_depth", "scratch", "has_exposure")

    def __init__(self, d_render, d_depth, scratch, has_exposure):
        self.d_render, self.d_depth, self.scratch, self.has_exposure = d_render, d_depth, scratch, has_exposure

    @property
    def loss(self) -> torch.Tensor:                    # device scalar, valid once the two kernels have run
        return self.scratch[_LOSS]

    @property
    def d_exposure_a(self):
        return self.scratch[_DAB:_DAB + 1] if self.has_exposure else None

    @property
    def d_exposure_b(self):
        return self.scratch[_DAB + 1:_DAB + 2] if self.has_exposure else None

    def backward(self, render_image, render_depth, viewpoint=None, accumulate=False):
        """Drive the rasteriser's backward with these gradients and hand the exposure gradients to the viewpoint."""
        torch.autograd.backward([render_image, render_depth], [self.d_render, self.d_depth])
        if viewpoint is not None and self.has_exposure:
            for p, g in ((viewpoint.exposure_a, self.d_exposure_a), (viewpoint.exposure_b, self.d_exposure_b)):
                p.grad = g if (p.grad is None or not accumulate) else p.grad + g


@torch.no_grad()
def loss_grads(render_image, render_depth, render_opacity, viewpoint, tracking: bool, init: bool = False,
               lambda_depth: float = 0.9, invert_depth: bool = False) -> LossGrads:
    """``get_loss_tracking`` / ``get_loss_mapping`` (/root/reference/utils/slam_utils.py:58-146) as VALUE + GRADIENTS in two
    launches, for loops that call the rasteriser's backward themselves: no autograd node for the scalar, hence no finalize
    kernel, no ones-fill and no loss-summing adds (``loss.backward()`` on the fused autograd loss costs four launches per
    render).  Same numbers as ``_FusedLoss`` forward + backward with grad_output = 1."""
    lib = _lib.load()
    render = _f32(render_image.detach(), "render_image")
    depth = _f32(render_depth.detach(), "render_depth")
    H, W = render.shape[-2:]
    dev = render.device
    opac = _f32(render_opacity.detach(), "render_opacity") if tracking else None
    gt_rgb, gt_depth = _f32(viewpoint.rgb, "viewpoint.rgb"), _f32(viewpoint.depth, "viewpoint.depth")
    a = None if init else _f32(viewpoint.exposure_a.detach(), "exposure_a")
    b = None if init else _f32(viewpoint.exposure_b.detach(), "exposure_b")
    mask = _u8(viewpoint.mask)
    gm = _u8(viewpoint.grad_mask) if tracking else None
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with _device_guard(dev):
        scratch = torch.empty(lib.mgs_loss_scratch_bytes() // 4, dtype=torch.float32, device=dev)
        d_render, d_depth = torch.empty_like(render), torch.empty_like(depth)
        mode = (_TRACKING if tracking else 0) | (_INVERT if invert_depth else 0)
        _lib.check(lib.mgs_loss_grads(W, H, mode, int(init), 0.9 if tracking else float(lambda_depth), p(render),
                                      p(depth), p(opac), p(gt_rgb), p(gt_depth), p(mask), p(gm), p(a), p(b), p(scratch),
                                      p(d_render), p(d_depth), _stream()), "mgs_loss_grads")
    return LossGrads(d_render, d_depth, scratch, not init)
