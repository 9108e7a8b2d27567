"""TEST INFRASTRUCTURE (checker for monogs_amd.fused_losses; the product package does not import it).

Photometric / geometric losses on the caller's side of the boundary, as MonoGS computes them.
They define the upstream gradients dL/dcolor, dL/ddepth the rasteriser's backward receives.

Mirrors /root/reference/utils/slam_utils.py:58-98 (tracking) and :101-146 (mapping); checked against
values and gradients produced by the reference's own functions (tests/golden/losses.npz).
"""
from __future__ import annotations

import torch


def get_loss_mapping(render_image, render_depth, viewpoint, init=False, invert_depth=False, lambda_depth=0.9):
    gt_rgb = viewpoint.rgb.permute(1, 2, 0)
    gt_mask = viewpoint.mask
    gt_depth = viewpoint.depth[None]
    rgb = render_image if init else torch.exp(viewpoint.exposure_a) * render_image + viewpoint.exposure_b
    rgb = rgb.permute(1, 2, 0)
    l1_rgb = (torch.abs(rgb[gt_mask] - gt_rgb[gt_mask]) if gt_mask is not None else torch.abs(rgb - gt_rgb)).mean()
    valid = gt_depth > 0
    if invert_depth:
        l1_depth = torch.abs(1 / render_depth[valid] - 1 / gt_depth[valid]).mean()
    else:
        l1_depth = torch.abs(render_depth[valid] - gt_depth[valid]).mean()
    return lambda_depth * l1_rgb + (1 - lambda_depth) * l1_depth


def get_loss_tracking(render_image, render_depth, render_opacity, viewpoint, invert_depth=False):
    gt_rgb = viewpoint.rgb
    gt_mask = viewpoint.mask
    gt_depth = viewpoint.depth[None]
    opacity_mask = render_opacity > 0.99
    rgb = torch.exp(viewpoint.exposure_a) * render_image + viewpoint.exposure_b
    rgb_mask = gt_mask * viewpoint.grad_mask * opacity_mask
    l1_rgb = (render_opacity * torch.abs(rgb * rgb_mask - gt_rgb * rgb_mask).mean()).mean()
    depth_mask = (gt_depth > 0) * opacity_mask
    if depth_mask.any():
        if invert_depth:
            l1_depth = torch.abs(1 / (render_depth[depth_mask] + 1e-6) - 1 / (gt_depth[depth_mask] + 1e-6)).mean()
        else:
            l1_depth = torch.abs(render_depth[depth_mask] - gt_depth[depth_mask]).mean()
    else:
        l1_depth = torch.tensor(0.0, device=render_depth.device, dtype=render_depth.dtype)
    return 0.5 * l1_rgb + l1_depth


@torch.no_grad()
def get_median_depth(depth, mask=None, return_std=False):
    """/root/reference/utils/slam_utils.py:149-157 (median of the valid depths; optionally their std and the mask)."""
    valid = depth > 0
    if mask is not None:
        valid = torch.logical_and(valid, mask)
    valid_depth = depth[valid]
    if return_std:
        return valid_depth.median(), valid_depth.std(), valid
    return valid_depth.median()
