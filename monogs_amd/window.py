"""Mapping-window sharding: independent keyframe renders, one (or a few) per GPU, and one
all-reduce of the Gaussian gradients per mapping iteration.

What is sharded: the inner loop of ``Mapper.optimize_map`` -- every keyframe of the window is
rendered against the SAME Gaussians and the losses are summed before a single backward
(/root/reference/utils/slam_mapper.py:273-324,394).  So rank r renders keyframes
``k % world == r``; the Gaussian parameters are replicated; their gradients are summed across
ranks with one collective over a single flat bucket (12 floats per Gaussian for the isotropic
map: xyz 3, rgb 3, opacity 1, scale 1, rotation 4; /root/reference/gaussian_splatting/scene/
gaussian_model.py:405-436); per-keyframe pose / exposure parameters stay on the owning rank.
The densification statistics need the per-keyframe norm of the screen-space gradient summed
over keyframes (gaussian_model.py:888-892), which is NOT the norm of the summed gradient, so it
travels as two extra columns of the same bucket; max_radii_2d needs a MAX reduction.

Backend: ``nccl`` (= RCCL over xGMI on MI355X) on GPU, ``gloo`` on CPU for the tests.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_keyframes(n_keyframes: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: keyframe k belongs to rank k % world."""
    return [k for k in range(n_keyframes) if k % world == rank]


# Above this many bytes the gradients are reduced in place, tensor by tensor (5 large collectives),
# instead of being packed into one bucket: at 2 M Gaussians the pack + unpack copies (2 x 96 MB) cost
# more than four extra collective launches; small SLAM maps (a few MB) keep the single latency-bound message.
PER_TENSOR_BYTES = 32 << 20


class GradBucket:
    """One flat [P, C] float32 buffer holding every Gaussian gradient column, reduced in one call
    (or, for large maps, the gradient tensors themselves reduced in place)."""

    def __init__(self, params: Sequence[torch.Tensor], extra_cols: int = 0, per_tensor: Optional[bool] = None):
        self.params = list(params)
        P = self.params[0].shape[0]
        self.widths = [int(p.numel() // P) for p in self.params]
        self.extra_cols = extra_cols
        total_bytes = 4 * P * (sum(self.widths) + extra_cols)
        self.per_tensor = (total_bytes > PER_TENSOR_BYTES) if per_tensor is None else per_tensor
        cols = extra_cols if self.per_tensor else sum(self.widths) + extra_cols
        self.buf = torch.zeros(P, cols, dtype=torch.float32, device=self.params[0].device)

    def pack(self, extra: Optional[torch.Tensor] = None):
        c = 0
        P = self.buf.shape[0]
        if self.per_tensor:
            for p in self.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            if self.extra_cols:
                self.buf.zero_() if extra is None else self.buf.copy_(extra)
            return
        for p, w in zip(self.params, self.widths):
            g = p.grad
            if g is None:
                self.buf[:, c:c + w].zero_()
            else:
                self.buf[:, c:c + w].copy_(g.reshape(P, w))
            c += w
        if self.extra_cols:
            if extra is None:
                self.buf[:, c:].zero_()
            else:
                self.buf[:, c:].copy_(extra)

    def unpack(self) -> Optional[torch.Tensor]:
        c = 0
        if self.per_tensor:
            return self.buf if self.extra_cols else None
        for p, w in zip(self.params, self.widths):
            g = self.buf[:, c:c + w].reshape(p.shape)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            c += w
        return self.buf[:, c:] if self.extra_cols else None

    def all_reduce(self, group=None, async_op: bool = False):
        if not (dist.is_available() and dist.is_initialized()):
            return None
        if self.per_tensor:
            tensors = [p.grad for p in self.params] + ([self.buf] if self.extra_cols else [])
            works = None
            if dist.get_backend(group) == "nccl" and hasattr(dist, "_coalescing_manager"):
                try:    # one RCCL group launch for all tensors (ncclGroupStart/End), no packing copies
                    with dist._coalescing_manager(group, device=tensors[0].device, async_ops=True) as cm:
                        for t in tensors:
                            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                    works = [cm]
                except Exception:
                    works = None
            if works is None:
                works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in tensors]
            if async_op:
                return works
            for w in works:
                w.wait()
            return None
        return dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def allreduce_window_grads(params: Sequence[torch.Tensor], viewspace_grad_norm: Optional[torch.Tensor] = None,
                           visible_count: Optional[torch.Tensor] = None, max_radii: Optional[torch.Tensor] = None,
                           group=None, bucket: Optional[GradBucket] = None):
    """Sum the Gaussian gradients (and optional densification statistics) over the ranks that
    rendered the window's keyframes.  Returns (bucket, grad_norm_sum, visible_sum, max_radii)."""
    n_extra = (viewspace_grad_norm is not None) + (visible_count is not None)
    if bucket is None:
        bucket = GradBucket(params, extra_cols=n_extra)
    extra = None
    if n_extra:
        cols = [t.reshape(-1, 1).to(torch.float32) for t in (viewspace_grad_norm, visible_count) if t is not None]
        extra = torch.cat(cols, dim=1)
    bucket.pack(extra)
    bucket.all_reduce(group)
    ex = bucket.unpack()
    gn = vs = None
    if ex is not None:
        i = 0
        if viewspace_grad_norm is not None:
            gn = ex[:, i]
            i += 1
        if visible_count is not None:
            vs = ex[:, i]
    if max_radii is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(max_radii, op=dist.ReduceOp.MAX, group=group)
    return bucket, gn, vs, max_radii
