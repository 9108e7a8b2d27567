"""``render()`` -- the seam MonoGS's tracker, mapper and viewer call.

Same signature, argument meaning, return dict and error behaviour as
/root/reference/gaussian_splatting/gaussian_renderer/__init__.py:26-168.  The camera arguments are
duck-typed: ``viewpoint_camera`` needs ``world_view_transform``, ``camera_center``,
``cam_rot_delta``, ``cam_trans_delta``; ``cam_intrinsics`` needs ``FoVx``, ``FoVy``, ``height``,
``width``, ``projection_matrix`` -- the reference's own CameraExtrinsics / CameraIntrinsics
(/root/reference/utils/camera_utils.py:8-79,82-221) satisfy this, as do the light-weight
stand-ins in ``monogs_amd.slam_harness``.
"""
from __future__ import annotations

import math

import torch

from .camera import cached_camera_tensors
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def _camera_center(world_view: torch.Tensor) -> torch.Tensor:
    """Camera centre -R^T t read off the transposed world->view matrix.  Same value as the reference's
    ``camera_center`` property (``world_view_transform.inverse()[3, :3]``,
    /root/reference/utils/camera_utils.py:176-178) without a 4x4 LU inverse per render; it only feeds the
    SH view direction, which this fork never uses (colours are precomputed)."""
    return -(world_view[:3, :3] @ world_view[3, :3])


def render(viewpoint_camera, cam_intrinsics, means, rotations, scales, opacity, features, bg_color,
           scaling_modifier=1.0, override_color=None, mask=None):
    if means.shape[0] == 0:
        return None
    # zero tensor whose .grad receives the screen-space mean gradients
    # (a leaf: the reference adds `+ 0` and calls retain_grad(); `.grad` is populated either way, with one kernel less)
    screenspace_points = torch.zeros_like(means, dtype=means.dtype, requires_grad=True, device=means.device)

    tanfovx = math.tan(cam_intrinsics.FoVx * 0.5)
    tanfovy = math.tan(cam_intrinsics.FoVy * 0.5)
    projection_matrix = cam_intrinsics.projection_matrix
    R, T = getattr(viewpoint_camera, "R", None), getattr(viewpoint_camera, "T", None)
    if (torch.is_tensor(R) and torch.is_tensor(T) and R.is_cuda and T.is_cuda and R.shape == (3, 3) and T.shape == (3,)
            and projection_matrix.is_cuda):
        # the reference's camera objects carry R, T (utils/camera_utils.py:82-221): all three camera tensors in one launch
        # (remembered on the camera object until R / T are replaced; the fused pose step keeps the cache current in place)
        world_view, full_proj, campos = cached_camera_tensors(viewpoint_camera, R, T, projection_matrix)
    else:
        world_view = viewpoint_camera.world_view_transform
        full_proj = (world_view.unsqueeze(0).bmm(projection_matrix.unsqueeze(0))).squeeze(0)
        campos = _camera_center(world_view)

    raster_settings = GaussianRasterizationSettings(
        image_height=int(cam_intrinsics.height), image_width=int(cam_intrinsics.width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=world_view, projmatrix=full_proj, projmatrix_raw=projection_matrix,
        sh_degree=0, campos=campos, prefiltered=False, debug=False)

    # isotropic map: the reference expands with scales.repeat(1, 3) here (gaussian_renderer/__init__.py:101-104); the
    # kernels take the [P,1] tensor as it is (mgs_camera.scale_dim = 1) and sum the three gradients themselves
    colors = features if override_color is None else override_color

    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    sel = (lambda t: t[mask]) if mask is not None else (lambda t: t)
    # The reference's `mask` branch unpacks 4 outputs and then reads an undefined n_touched
    # (gaussian_renderer/__init__.py:131-143,167); no caller passes a mask.  Here the branch works.
    rendered_image, radii, depth, opacity_img, n_touched = rasterizer(
        means3D=sel(means), means2D=sel(screenspace_points), shs=None, colors_precomp=sel(colors),
        opacities=sel(opacity), scales=sel(scales), rotations=sel(rotations), cov3D_precomp=None,
        theta=viewpoint_camera.cam_rot_delta, rho=viewpoint_camera.cam_trans_delta)

    return {
        "render": rendered_image,
        "viewspace_points": screenspace_points,
        "visibility_filter": radii > 0,
        "radii": radii,
        "depth": depth,
        "opacity": opacity_img,
        "n_touched": n_touched,
    }
