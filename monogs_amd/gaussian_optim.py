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
from .rasterizer import _stream, _device_guard


class GaussianAdam:
    def __init__(self, params: Sequence[torch.Tensor], lrs: Sequence[float], betas=(0.9, 0.999), eps=1e-15):
        assert 1 <= len(params) <= 8 and len(params) == len(lrs)
        self.params = list(params)
        self.lrs = [float(x) for x in lrs]
        self.betas, self.eps = betas, float(eps)
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.t_dev = torch.zeros(len(self.params), dtype=torch.int32, device=self.params[0].device)   # per-tensor step counts
        self.lr_dev = None          # device float[n]: learning rates read by the kernel instead of ``lrs`` (device_lrs())

    # ---- optimiser-state surgery (map growth / pruning) -----------------------------------------------------
    # What GaussianModel does to torch.optim.Adam's state when the map changes size
    # (/root/reference/gaussian_splatting/scene/gaussian_model.py:642-776): the moments of surviving Gaussians are
    # carried over exactly, new Gaussians start from zero moments, a replaced tensor restarts from zero moments.  The
    # step count is untouched in all three cases (torch keeps ``state["step"]`` across the surgery as well); it is kept
    # PER TENSOR on the device (``t_dev``), as torch keeps one per parameter -- a tensor whose gradient is None skips its
    # step and its count, exactly like torch.  Each method returns the new leaf tensors (requires_grad) that replace
    # ``params``.
    @torch.no_grad()
    def extend(self, new_tensors: Sequence[torch.Tensor]):
        """``cat_tensors_to_optimizer`` (gaussian_model.py:709-743): append rows, zero moments for them."""
        assert len(new_tensors) == len(self.params)
        for i, ext in enumerate(new_tensors):
            ext = ext.detach().to(self.params[i].dtype)
            assert ext.shape[1:] == self.params[i].shape[1:], (i, ext.shape, self.params[i].shape)
            self.params[i] = torch.cat((self.params[i].detach(), ext), 0).requires_grad_(True)
            self.exp_avg[i] = torch.cat((self.exp_avg[i], torch.zeros_like(ext)), 0)
            self.exp_avg_sq[i] = torch.cat((self.exp_avg_sq[i], torch.zeros_like(ext)), 0)
        return list(self.params)

    @torch.no_grad()
    def prune(self, keep_mask: torch.Tensor):
        """``_prune_optimizer`` (gaussian_model.py:658-680): keep the rows where ``keep_mask`` is True."""
        keep_mask = keep_mask.to(self.params[0].device).bool().reshape(-1)
        assert keep_mask.shape[0] == self.params[0].shape[0]
        for i in range(len(self.params)):
            self.params[i] = self.params[i].detach()[keep_mask].requires_grad_(True)
            self.exp_avg[i] = self.exp_avg[i][keep_mask]
            self.exp_avg_sq[i] = self.exp_avg_sq[i][keep_mask]
        return list(self.params)

    @torch.no_grad()
    def replace(self, index: int, tensor: torch.Tensor):
        """``replace_tensor_to_optimizer`` (gaussian_model.py:642-656): new values, zero moments (opacity reset)."""
        assert tensor.shape == self.params[index].shape
        self.params[index] = tensor.detach().clone().requires_grad_(True)
        self.exp_avg[index] = torch.zeros_like(self.params[index])
        self.exp_avg_sq[index] = torch.zeros_like(self.params[index])
        return self.params[index]

    def set_lr(self, index: int, lr: float):
        """Learning rate of tensor ``index`` for the following steps (``update_learning_rate`` sets the xyz group's)."""
        self.lrs[index] = float(lr)
        if self.lr_dev is not None:
            self.lr_dev[index] = float(lr)

    def device_lrs(self, on: bool = True):
        """Keep the learning rates in device memory (a captured mapping iteration steps the xyz schedule there,
        ``mgs_lr_schedule_step``).  Returns the device tensor; ``sync_lrs_from_device`` copies it back into ``lrs``."""
        if not on:
            self.lr_dev = None
        elif self.lr_dev is None:
            self.lr_dev = torch.tensor(self.lrs, dtype=torch.float32, device=self.params[0].device)
        return self.lr_dev

    def sync_lrs_from_device(self):
        if self.lr_dev is not None:
            self.lrs = [float(x) for x in self.lr_dev.tolist()]

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
        if all(g is None for g in grads):       # nothing to do (torch.optim.Adam: every parameter skipped)
            return
        tab = lambda ts: vp(*[None if t is None else t.data_ptr() for t in ts])  # noqa: E731
        numel = (C.c_uint64 * n)(*[p.numel() for p in self.params])
        lr = (C.c_float * n)(*self.lrs)
        with _device_guard(self.params[0].device):
            _lib.check(lib.mgs_adam_step(n, tab(self.params), tab(grads), tab(self.exp_avg), tab(self.exp_avg_sq), numel,
                                         lr, self.betas[0], self.betas[1], self.eps, 0, self.t_dev.data_ptr(),
                                         None if self.lr_dev is None else self.lr_dev.data_ptr(), _stream()),
                       "mgs_adam_step")


@torch.no_grad()
def expon_lr(step: int, lr_init: float, lr_final: float, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0,
             max_steps: int = 1000000) -> float:
    """Position learning-rate schedule of ``GaussianModel.update_learning_rate``
    (/root/reference/gaussian_splatting/scene/gaussian_model.py:451-465 -> ``helper`` in
    /root/reference/gaussian_splatting/utils/general_utils.py:79-94): log-linear interpolation from ``lr_init`` to
    ``lr_final`` over ``max_steps`` with an optional eased-in delay factor.  Checked against the reference's own
    outputs (tests/golden/lr_schedule.npz)."""
    import math
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    delay = 1.0
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1.0 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
    t = min(max(step / max_steps, 0.0), 1.0)
    return delay * math.exp(math.log(lr_init) * (1.0 - t) + math.log(lr_final) * t)


def add_densification_stats(viewspace_grad: torch.Tensor, radii: torch.Tensor, xyz_gradient_accum=None, denom=None,
                            max_radii_2d=None):
    """In-place update of the three statistics for one rendered keyframe (visible = radii > 0)."""
    lib = _lib.load()
    P = radii.shape[0]
    g = viewspace_grad.contiguous()
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with _device_guard(radii.device):
        _lib.check(lib.mgs_densify_stats(P, g.data_ptr(), radii.contiguous().data_ptr(), p(xyz_gradient_accum), p(denom),
                                         p(max_radii_2d), _stream()), "mgs_densify_stats")


def window_stats(grads2d, radii, n_touched, grad_norm, visible, max_radii, accumulate: bool, bits=None):
    """``mgs_window_stats``: the statistics of one mapping iteration over the keyframes this rank rendered, one launch
    (/root/reference/utils/slam_mapper.py:400-404,453-460).  ``grads2d[k]`` / ``radii[k]`` / ``n_touched[k]``: per keyframe
    (a gradient may be None); the three [P] float arrays are updated in place (``accumulate``) or overwritten with this
    rank's share; ``bits``: int64 [rows >= K, ceil(P/64)] receiving the packed ``n_touched > 0`` bits of keyframe k in row k."""
    K = len(radii)
    if K == 0:
        if not accumulate:
            grad_norm.zero_(); visible.zero_(); max_radii.zero_()
        return
    lib = _lib.load()
    P = int(radii[0].shape[0])
    keep = [None if g is None else g.contiguous() for g in grads2d] + [r.contiguous() for r in radii] + \
           [t.contiguous() for t in n_touched]
    arr = lambda ts: (C.c_void_p * K)(*[None if t is None else t.data_ptr() for t in ts])  # noqa: E731
    if bits is not None:
        assert bits.dtype == torch.int64 and bits.is_contiguous() and bits.shape[0] >= K and bits.shape[1] == (P + 63) // 64
    with _device_guard(radii[0].device):
        _lib.check(lib.mgs_window_stats(P, K, arr(keep[:K]), arr(keep[K:2 * K]), arr(keep[2 * K:]), grad_norm.data_ptr(),
                                        visible.data_ptr(), max_radii.data_ptr(), 1 if accumulate else 0,
                                        None if bits is None else bits.data_ptr(), _stream()), "mgs_window_stats")


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rot_raw, scale_raw, opacity_raw):
        lib = _lib.load()
        rot_raw, scale_raw, opacity_raw = rot_raw.detach().contiguous(), scale_raw.detach().contiguous(), opacity_raw.detach().contiguous()
        P, sd = rot_raw.shape[0], scale_raw.shape[1]
        dev = rot_raw.device
        rot = torch.empty(P, 4, device=dev)
        scales3 = torch.empty(P, 3, device=dev)
        opac = torch.empty(P, 1, device=dev)
        with _device_guard(dev):
            _lib.check(lib.mgs_activate_forward(P, sd, rot_raw.data_ptr(), scale_raw.data_ptr(), opacity_raw.data_ptr(),
                                                rot.data_ptr(), scales3.data_ptr(), opac.data_ptr(), _stream()),
                       "mgs_activate_forward")
        ctx.save_for_backward(rot_raw, scales3, opac)
        ctx.sd = sd
        return rot, scales3, opac

    @staticmethod
    def backward(ctx, g_rot, g_scales3, g_opac):
        lib = _lib.load()
        rot_raw, scales3, opac = ctx.saved_tensors
        P, dev = rot_raw.shape[0], rot_raw.device
        need = ctx.needs_input_grad
        d_rot = torch.empty(P, 4, device=dev) if need[0] else None
        d_scale = torch.empty(P, ctx.sd, device=dev) if need[1] else None
        d_opac = torch.empty(P, 1, device=dev) if need[2] else None
        p = lambda t: None if t is None else t.contiguous().data_ptr()  # noqa: E731
        keep = [t.contiguous() if t is not None else None for t in (g_rot, g_scales3, g_opac)]
        with _device_guard(dev):
            _lib.check(lib.mgs_activate_backward(P, ctx.sd, rot_raw.data_ptr(), scales3.data_ptr(), opac.data_ptr(),
                                                 p(keep[0]), p(keep[1]), p(keep[2]), p(d_rot), p(d_scale), p(d_opac),
                                                 _stream()), "mgs_activate_backward")
        return d_rot, d_scale, d_opac


def sum_buffers(srcs, out=None):
    """``out = srcs[0] + srcs[1] + ...`` (contiguous float32 device tensors of one size) in ONE launch (``mgs_sum_buffers``;
    more than 16 sources: groups of 16, the running sum first).  The order of the adds is fixed."""
    lib = _lib.load()
    srcs = list(srcs)
    if out is None:
        out = torch.empty_like(srcs[0])
    grp, rest = srcs[:16], srcs[16:]
    with _device_guard(out.device):
        while True:
            arr = (C.c_void_p * len(grp))(*[t.data_ptr() for t in grp])
            _lib.check(lib.mgs_sum_buffers(len(grp), arr, out.data_ptr(), out.numel(), _stream()), "mgs_sum_buffers")
            if not rest:
                break
            grp, rest = [out] + rest[:15], rest[15:]
    return out


class _FanOut(torch.autograd.Function):
    """n aliases of each of m tensors; backward adds the n incoming gradients of every tensor in one launch."""

    @staticmethod
    def forward(ctx, n, out, *tensors):
        ctx.n, ctx.m, ctx.out = int(n), len(tensors), out
        ctx.set_materialize_grads(False)
        return tuple(t.view_as(t) for _ in range(int(n)) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        n, m = ctx.n, ctx.m
        per = [grads[k * m:(k + 1) * m] for k in range(n)]
        ok = lambda g: g is not None and g.dtype == torch.float32 and g.is_contiguous()  # noqa: E731

        def flat_of(gs):            # the m gradients of one consumer as ONE buffer, if they lie back to back
            if not all(ok(g) for g in gs):
                return None
            end = gs[0].data_ptr()
            for g in gs:
                if g.data_ptr() != end:
                    return None
                end += g.numel() * 4
            st = gs[0].untyped_storage()
            if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in gs):
                return None
            total = sum(g.numel() for g in gs)
            return torch.as_strided(gs[0], (total,), (1,), gs[0].storage_offset())

        live = [gs for gs in per if any(g is not None for g in gs)]
        flats = [flat_of(gs) for gs in live]
        if live and all(f is not None for f in flats) and len({f.numel() for f in flats}) == 1:
            # (the rasteriser's backward carves its map gradients out of one allocation, in argument order)
            if ctx.out is not None and ctx.out.numel() == flats[0].numel():
                total = sum_buffers(flats, out=ctx.out)          # the caller's bucket (one source: a copy into it)
            else:
                total = flats[0] if len(flats) == 1 else sum_buffers(flats)
            out, o = [], 0
            for g in live[0]:
                out.append(total[o:o + g.numel()].view(g.shape))
                o += g.numel()
            return (None, None, *out)
        out = []
        for j in range(m):
            gj = [gs[j] for gs in per if gs[j] is not None]
            if not gj:
                out.append(None)
            elif len(gj) == 1:
                out.append(gj[0])
            elif all(ok(g) for g in gj):
                out.append(sum_buffers(gj))
            else:
                acc = gj[0]
                for g in gj[1:]:
                    acc = acc + g
                out.append(acc)
        if ctx.out is not None:
            raise RuntimeError("fan_out(out=...): the consumers' gradients do not lie back to back in one buffer each")
        return (None, None, *out)


def fan_out(n: int, *tensors, out=None):
    """``n`` tuples of aliases of ``tensors`` for ``n`` consumers (the keyframe renders of a mapping window).  Handing every
    render its own aliases makes the autograd engine deliver the n gradients of each tensor to ONE node, which adds them in
    one launch -- all m tensors at once when a consumer returns its gradients back to back in one buffer, as the
    rasteriser's backward does for (means3D, colours, opacities, scales, rotations) in that order -- instead of n - 1
    pairwise adds per tensor.  ``out``: a flat float32 buffer of the summed size that receives the sum (the exchange
    bucket of a mapping window: the gradients of all keyframes land where the all-reduce reads them)."""
    flat = _FanOut.apply(n, out, *tensors)
    m = len(tensors)
    return [flat[k * m:(k + 1) * m] for k in range(n)]


def activate(rot_raw: torch.Tensor, scale_raw: torch.Tensor, opacity_raw: torch.Tensor):
    """(normalize(rot_raw), exp(scale_raw) expanded to [P,3], sigmoid(opacity_raw)) in one launch, differentiable."""
    return _Activate.apply(rot_raw, scale_raw, opacity_raw)
