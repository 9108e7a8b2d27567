"""Keyframe back-projection: new Gaussians from an RGB-D keyframe.

Mirror of ``GaussianModel.create_viewpoint_pcd``
(/root/reference/gaussian_splatting/scene/gaussian_model.py:121-319): same arguments, same six results
``(points_3d, features, scales, rots, opacities, points_ids)``, same densification mask

    depth >= 1e-3  and  ( O(p) < 0.5  or  ( D_gt(p) < D(p)  and  |D_gt - D|(p) > 50 * median|D_gt - D| ) )

same down-sampling ratio (1/32 at initialisation, 1/64 afterwards), same scale rule
``log(sqrt(clamp_min(distCUDA2(points), 1e-7) * point_size))`` with ``point_size = min(0.05, 0.01 * median(depth))``.

What changes is where the work runs.  The reference draws ``torch.randperm(num_points)`` on the CPU and indexes the
device tensors with it (a host permutation of ~300 k elements and a host->device copy per keyframe) and reads two
medians back through Python ``min``; here the subset is drawn on the device (uniform without replacement: the
``k`` smallest of one uniform variate per candidate) and the per-point gather / exposure / unprojection /
camera->world chain is one HIP launch (``mgs_backproject``).  The only host read-back is the candidate count, which
sizes the result.  The random stream therefore differs from the reference's; pass ``random_indices`` (what the
reference's ``torch.randperm(num_points)[:k]`` would be) to reproduce a given subset exactly -- the parity test does.
The reference cannot be imported here (open3d is absent), so this file is checked against a plain PyTorch
restatement of the lines cited above: parity unpinned.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .knn import distCUDA2
from .rasterizer import _stream, _device_guard


def densification_mask(gt_depth: torch.Tensor, render_depth: Optional[torch.Tensor], render_opacity: Optional[torch.Tensor],
                       init: bool) -> torch.Tensor:
    """bool[W*H] in the reference's flattening order (x outer, y inner)."""
    d = gt_depth.t().reshape(-1)
    mask = d >= 1e-3
    if init:
        return mask
    rd = render_depth.reshape(gt_depth.shape).t().reshape(-1)
    low_opacity = (render_opacity.reshape(gt_depth.shape).t().reshape(-1) < 0.5) if render_opacity is not None \
        else torch.ones_like(mask)
    err = (d - rd).abs()
    in_front = (d < rd) & (err > 50 * err.median())
    return mask & (low_opacity | in_front)


@torch.no_grad()
def create_viewpoint_pcd(viewpoint, cam_intrinsics, render_depth=None, render_opacity=None, init=False,
                         isotropic=True, random_indices: Optional[torch.Tensor] = None,
                         generator: Optional[torch.Generator] = None, downsample_factor: Optional[int] = None,
                         point_size: float = 0.01, point_size_max: float = 0.05,
                         knn_against: Optional[torch.Tensor] = None):
    """``downsample_factor`` / ``point_size`` / ``point_size_max`` default to the values hard-coded in the reference
    (32 or 64, 0.01, 0.05: gaussian_model.py:166-178).

    ``knn_against`` (OPT-IN, changes results relative to the reference): the positions ``[M,3]`` of the Gaussians already in
    the map.  The reference sizes a new Gaussian from its 3 nearest neighbours among the NEW points only and leaves
    "TODO: should compute against all existing gaussians" (gaussian_model.py:293); with this argument the neighbours are
    searched in new + existing points (one Morton-box kNN over the concatenated cloud: ~3 ms at 2 M points), so a point
    that lands next to mapped geometry gets a scale that fits it instead of the spacing of the sparse new sample."""
    lib = _lib.load()
    rgb = viewpoint.rgb.to(torch.float32).contiguous()
    depth = viewpoint.depth.to(torch.float32).contiguous()
    dev = depth.device
    H, W = depth.shape
    seg = getattr(viewpoint, "segmentation", None)
    mask = densification_mask(depth, render_depth, render_opacity, init)
    cand = mask.nonzero().squeeze(1)                       # candidate pixels, reference order (one host read-back: its length)
    n = int(cand.numel())
    keep = int(n * (1.0 / (downsample_factor if downsample_factor else (32 if init else 64))))
    if random_indices is not None:
        pick = random_indices.to(dev)[:keep]
    else:                                                  # uniform subset without replacement, drawn on the device
        u = torch.rand(n, device=dev, generator=generator)
        pick = u.topk(keep, largest=False).indices if keep > 0 else torch.empty(0, dtype=torch.long, device=dev)
    sel = cand[pick].contiguous()
    N = int(sel.numel())
    point_size = torch.clamp_max(point_size * depth.median(), point_size_max)          # stays on the device
    pts = torch.empty(N, 3, device=dev)
    feat = torch.empty(N, 3, device=dev)
    ids = torch.empty(N, dtype=torch.int32, device=dev) if seg is not None else None
    seg32 = seg.to(torch.int32).contiguous() if seg is not None else None
    k = cam_intrinsics
    fx, fy, cx, cy = (float(getattr(k, a)) for a in ("fx", "fy", "cx", "cy"))
    R, T = viewpoint.R.to(torch.float32).contiguous(), viewpoint.T.to(torch.float32).contiguous()
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    ea = None if init else viewpoint.exposure_a.detach().to(torch.float32).contiguous()
    eb = None if init else viewpoint.exposure_b.detach().to(torch.float32).contiguous()
    with _device_guard(dev):
        _lib.check(lib.mgs_backproject(N, W, H, p(sel), p(rgb), p(depth), p(seg32), p(ea), p(eb), fx, fy, cx, cy, p(R), p(T),
                                       p(pts), p(feat), p(ids), _stream()), "mgs_backproject")
    if N > 0:
        if knn_against is not None and knn_against.shape[0] > 0:
            cloud = torch.cat([pts, knn_against.detach().to(pts.dtype).reshape(-1, 3)], 0).contiguous()
            d2 = distCUDA2(cloud)[:N]
        else:
            d2 = distCUDA2(pts)
        dist2 = torch.clamp_min(d2, 1e-7) * point_size
        scales = torch.log(torch.sqrt(dist2))[:, None]
    else:
        scales = torch.empty(0, 1, device=dev)
    if not isotropic:
        scales = scales.repeat(1, 3)
    rots = torch.zeros(N, 4, device=dev)
    rots[:, 0] = 1
    opacities = torch.zeros(N, 1, device=dev)              # inverse_sigmoid(0.5)
    return pts, feat, scales, rots, opacities, ids
