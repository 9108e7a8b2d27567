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

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_keyframes(n_keyframes: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: keyframe k belongs to rank k % world."""
    return [k for k in range(n_keyframes) if k % world == rank]


def _staged(t: torch.Tensor, group=None) -> bool:
    """gloo moves host memory only: device tensors are staged through the CPU (the 2-rank rehearsal of the sharded
    window on a single GPU, tests/test_gpu_window.py).  RCCL (`nccl`) reduces device memory in place."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_(t: torch.Tensor, op=None, group=None, async_op: bool = False):
    """In-place all-reduce of ``t`` on whatever backend the group runs (RCCL on GPUs, gloo on CPU / staged)."""
    op = dist.ReduceOp.SUM if op is None else op
    if _staged(t, group):
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=op, group=group, async_op=async_op)


def all_gather_into_(out: torch.Tensor, inp: torch.Tensor, group=None):
    if _staged(inp, group):
        ho = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(ho, inp.detach().cpu(), group=group)
        out.copy_(ho)
        return
    dist.all_gather_into_tensor(out, inp, group=group)


def pipelined_all_reduce(n_slots: int, produce, buffers: Sequence[torch.Tensor], group=None, overlap: bool = True,
                         n_owned: Optional[int] = None) -> None:
    """SUM-all-reduce ``buffers[j]`` for every slot j, where ``produce(j)`` queues the work that fills ``buffers[j]`` (one
    owned keyframe's render + backward).  ``overlap``: slot j's collective is issued right behind ``produce(j)``,
    asynchronously, so that it runs while ``produce(j + 1)`` executes (RCCL: on the communicator's stream, ordered behind
    the producer by an event); otherwise all collectives are issued after the last producer.  Same collectives, same data,
    same order either way -- the results are bit-identical, only the timing differs.  Returns when every collective has been
    waited for (the caller's stream then sees the reduced buffers).

    ``n_slots`` must be the SAME on every rank -- ``rows_per_rank(window, world)``, not the number of keyframes this rank
    happens to own: with ``k % world`` sharding a window that is not a multiple of the world size (a SLAM window growing from
    1 to 8 keyframes passes through every size) leaves some ranks one keyframe short, and ranks that disagree on the number or
    the sizes of their collectives hang or reduce garbage.  ``n_owned`` (default: all) is how many of the slots this rank fills;
    the others contribute zeros (the in-place all-reduce leaves the other ranks' sum in them, so they are cleared each time)."""
    n_owned = n_slots if n_owned is None else int(n_owned)
    assert 0 <= n_owned <= n_slots <= len(buffers)
    works = []
    for j in range(n_slots):
        if j < n_owned:
            produce(j)
        else:
            buffers[j].zero_()
        if overlap:
            works.append(all_reduce_(buffers[j], group=group, async_op=True))
    if not overlap:
        works = [all_reduce_(b, group=group, async_op=True) for b in buffers[:n_slots]]
    for w in works:
        if w is not None:
            w.wait()


# Above this many bytes the gradients are reduced in place, tensor by tensor (5 large collectives),
# instead of being packed into one bucket: at 2 M Gaussians the pack + unpack copies (2 x 96 MB) cost
# more than four extra collective launches; small SLAM maps (a few MB) keep the single latency-bound message.
PER_TENSOR_BYTES = 32 << 20


def flat_view(tensors: Sequence[torch.Tensor]) -> Optional[torch.Tensor]:
    """One 1-D tensor that aliases ALL of ``tensors`` if they are contiguous, of one dtype, and laid out back to back in
    one storage (as the rasteriser's backward allocates the map gradients); else None."""
    ts = [t for t in tensors if t is not None]
    if not ts or any((not t.is_contiguous()) or t.dtype != ts[0].dtype or t.device != ts[0].device for t in ts):
        return None
    base = ts[0].untyped_storage().data_ptr()
    nxt = ts[0].data_ptr()
    for t in ts:
        if t.untyped_storage().data_ptr() != base or t.data_ptr() != nxt:
            return None
        nxt += t.numel() * t.element_size()
    total = sum(t.numel() for t in ts)
    return ts[0].as_strided((total,), (1,), ts[0].storage_offset())


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
        self.last_collectives = 1            # collectives issued by the last all_reduce()

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
        """SUM over the ranks.  Per-tensor mode issues one asynchronous collective per gradient tensor (public API
        only; RCCL runs them back to back on its own stream) and waits for all of them unless ``async_op``."""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        if self.per_tensor:
            grads = [p.grad for p in self.params]
            flat = flat_view(grads)             # back-to-back gradients (the rasteriser's own allocation): ONE collective
            tensors = ([flat] if flat is not None else grads) + ([self.buf] if self.extra_cols else [])
            self.last_collectives = len(tensors)
            works = [all_reduce_(t, group=group, async_op=True) for t in tensors]
            works = [w for w in works if w is not None]
            if async_op:
                return works
            for w in works:
                w.wait()
            return None
        return all_reduce_(self.buf, group=group, async_op=async_op)


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
        all_reduce_(max_radii, op=dist.ReduceOp.MAX, group=group)
    return bucket, gn, vs, max_radii


# ---- the small exchanges that keep map management identical on every rank -------------------
def _world(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def _rank(group=None) -> int:
    return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def rows_per_rank(n_keyframes: int, world: int) -> int:
    return (n_keyframes + world - 1) // world


def pack_bits(mask: torch.Tensor) -> torch.Tensor:
    """bool[..., P] -> uint8[..., ceil(P/8)] (bit i of byte j = element 8j+i)."""
    P = mask.shape[-1]
    pad = (-P) % 8
    m = mask.to(torch.uint8)
    if pad:
        m = torch.nn.functional.pad(m, (0, pad))
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=mask.device)
    return (m.reshape(*mask.shape[:-1], -1, 8) * w).sum(-1, dtype=torch.uint8)


def unpack_bits(packed: torch.Tensor, P: int) -> torch.Tensor:
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=packed.device)
    bits = (packed.unsqueeze(-1) & w) != 0
    return bits.reshape(*packed.shape[:-1], -1)[..., :P]


def all_gather_visibility(local: Dict[int, torch.Tensor], n_keyframes: int, P: int, group=None) -> List[torch.Tensor]:
    """``occ_aware_visibility_dict[kf] = n_touched_kf > 0`` for EVERY keyframe of the window on every rank
    (/root/reference/utils/slam_mapper.py:400-404).  ``local`` maps the window positions this rank rendered to
    bool[P] (or the int n_touched itself); P bits per keyframe travel (all-gather of packed bytes).  Returns a list of
    bool[P], one per window position."""
    world, rank = _world(group), _rank(group)
    dev = next(iter(local.values())).device if local else torch.device("cpu")
    if world == 1:
        return [(local[k] > 0) if local[k].dtype != torch.bool else local[k] for k in range(n_keyframes)]
    rows = rows_per_rank(n_keyframes, world)
    nb = (P + 7) // 8
    mine = torch.zeros(rows, nb, dtype=torch.uint8, device=dev)
    for j, k in enumerate(shard_keyframes(n_keyframes, rank, world)):
        v = local[k]
        mine[j] = pack_bits(v if v.dtype == torch.bool else v > 0)
    allv = torch.empty(world * rows, nb, dtype=torch.uint8, device=dev)
    all_gather_into_(allv, mine, group=group)
    return [unpack_bits(allv[(k % world) * rows + k // world], P) for k in range(n_keyframes)]


POSE_FLOATS = 14      # R 9, T 3, exposure_a 1, exposure_b 1


def all_gather_poses(viewpoints: Sequence, group=None, force: bool = False) -> None:
    """After the window optimisation every rank needs the updated pose and exposure of EVERY keyframe before the map
    is handed to the front end (/root/reference/utils/slam_mapper.py:553-556).  Keyframe k is owned by rank k % world;
    the owners' values overwrite the stale copies on the other ranks, in place."""
    world, rank = _world(group), _rank(group)
    n = len(viewpoints)
    if (world == 1 and not (force and dist.is_available() and dist.is_initialized())) or n == 0:
        return
    rows = rows_per_rank(n, world)
    dev = viewpoints[0].R.device
    mine = torch.zeros(rows, POSE_FLOATS, device=dev)
    for j, k in enumerate(shard_keyframes(n, rank, world)):
        vp = viewpoints[k]
        mine[j] = torch.cat([vp.R.reshape(9), vp.T.reshape(3), vp.exposure_a.detach().reshape(1),
                             vp.exposure_b.detach().reshape(1)])
    allp = torch.empty(world * rows, POSE_FLOATS, device=dev)
    all_gather_into_(allp, mine, group=group)
    with torch.no_grad():
        for k, vp in enumerate(viewpoints):
            if k % world == rank:
                continue
            row = allp[(k % world) * rows + k // world]
            vp.R, vp.T = row[:9].reshape(3, 3).clone(), row[9:12].clone()
            vp.exposure_a.data.copy_(row[12:13])
            vp.exposure_b.data.copy_(row[13:14])


def split_generator(device, base_seed: int, iteration: int) -> torch.Generator:
    """The generator ``densify_and_split`` draws its offsets from (/root/reference/gaussian_splatting/scene/
    gaussian_model.py:793 uses the global RNG): seeded from (base_seed, mapping iteration) only, so replicas that hold
    bit-identical parameters split into bit-identical children on every rank -- no broadcast needed."""
    g = torch.Generator(device=device)
    g.manual_seed((int(base_seed) * 1000003 + int(iteration)) & 0x7FFFFFFFFFFFFFFF)
    return g


def replicas_in_sync(tensors: Sequence[torch.Tensor], group=None, force: bool = False) -> bool:
    """Cheap divergence check for the replicated map: a 64-bit wrap-around checksum of the raw bits of every tensor,
    compared across ranks with MIN / MAX reductions.  True when every rank holds the same bits."""
    if _world(group) == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return True
    sums = []
    for t in tensors:
        raw = t.detach().contiguous().view(torch.int32).to(torch.int64)
        sums.append(raw.sum() + 31 * t.shape[0])
    c = torch.stack(sums)
    lo, hi = c.clone(), c.clone()
    all_reduce_(lo, op=dist.ReduceOp.MIN, group=group)
    all_reduce_(hi, op=dist.ReduceOp.MAX, group=group)
    return bool((lo == hi).all().item())
