"""Deterministic synthetic scenes for the BASELINE.json configurations.

All tensors are drawn from a CPU ``torch.Generator`` seeded by the caller and then moved to
the requested device, so the oracle (CPU) and the HIP path (GPU) see identical bits.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch

from .camera import INTRINSICS, camera_matrices, se3_exp


class Scene(NamedTuple):
    means3D: torch.Tensor      # [P,3]
    scales: torch.Tensor       # [P,1] isotropic (or [P,3])
    rotations: torch.Tensor    # [P,4] unit (r,x,y,z)
    opacities: torch.Tensor    # [P,1]
    colors: torch.Tensor       # [P,3]
    bg: torch.Tensor           # [3]
    R: torch.Tensor            # [3,3] world->camera
    t: torch.Tensor            # [3]
    intr: dict
    grad_color: torch.Tensor   # [3,H,W] upstream dL/dcolor
    grad_depth: torch.Tensor   # [1,H,W] upstream dL/ddepth


# pose of SURVEY.md section 8d
_TAU = (0.1, -0.2, 0.3, 0.05, 0.02, -0.04)


def make_scene(P: int, intrinsics: str = "fr3_office", seed: int = 0, mean_radius_px: float = 6.0,
               anisotropic: bool = False, near_fraction: float = 0.01, bg=(0.0, 0.0, 0.0),
               pose_tau=_TAU, device="cpu", spread: float = 1.15) -> Scene:
    intr = dict(INTRINSICS[intrinsics]) if isinstance(intrinsics, str) else dict(intrinsics)
    W, H, fx, fy = intr["W"], intr["H"], intr["fx"], intr["fy"]
    g = torch.Generator(device="cpu").manual_seed(seed)
    U = lambda *s: torch.rand(*s, generator=g, dtype=torch.float32)          # noqa: E731
    N = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)         # noqa: E731

    T_cw = se3_exp(torch.tensor(pose_tau, dtype=torch.float32))
    R, t = T_cw[:3, :3].contiguous(), T_cw[:3, 3].contiguous()
    tanx, tany = W / (2 * fx), H / (2 * fy)

    z = 0.5 + 7.5 * U(P)
    n_near = int(P * near_fraction)
    if n_near:
        z[:n_near] = 0.05 + 0.14 * U(n_near)            # exercised by the z <= 0.2 near cull
    u, v = 2 * U(P) - 1, 2 * U(P) - 1
    pc = torch.stack([u * z * tanx * spread, v * z * tany * spread, z], dim=1)
    means3D = (pc - t[None, :]) @ R                      # R^T (pc - t), row form

    # isotropic scale s with projected sigma fx*s/z; radius ~= ceil(3*sqrt(sigma^2 + 0.3))
    sigma_px = max(0.3, (mean_radius_px - 0.5) / 3.0)
    s0 = sigma_px * 4.25 / fx * 0.40                     # 0.40: calibrates E[radius] under the log-uniform draw
    n_sc = 3 if anisotropic else 1
    scales = torch.exp(math.log(s0 / 3) + (math.log(3 * s0) - math.log(s0 / 3)) * U(P, n_sc))
    q = N(P, 4)
    rotations = q / q.norm(dim=1, keepdim=True)
    opacities = torch.sigmoid(1.5 * N(P, 1))
    colors = U(P, 3)
    grad_color = (2 * U(3, H, W) - 1) / (3 * H * W)
    grad_depth = (2 * U(1, H, W) - 1) / (H * W)
    to = lambda a: a.contiguous().to(device)              # noqa: E731
    return Scene(to(means3D), to(scales), to(rotations), to(opacities), to(colors),
                 to(torch.tensor(bg, dtype=torch.float32)), to(R), to(t), intr,
                 to(grad_color), to(grad_depth))


def scene_settings(scene: Scene, settings_cls, device: Optional[str] = None, scale_modifier=1.0,
                   sh_degree=0):
    """Build the 13-field settings tuple for ``scene`` with the given NamedTuple class
    (``GaussianRasterizationSettings`` or the oracle's ``OracleSettings``)."""
    device = device or scene.means3D.device
    i = scene.intr
    cm = camera_matrices(scene.R.cpu(), scene.t.cpu(), i["fx"], i["fy"], i["cx"], i["cy"], i["W"], i["H"])
    return settings_cls(
        image_height=i["H"], image_width=i["W"], tanfovx=cm.tanfovx, tanfovy=cm.tanfovy,
        bg=scene.bg.to(device), scale_modifier=scale_modifier,
        viewmatrix=cm.viewmatrix.to(device), projmatrix=cm.projmatrix.to(device),
        projmatrix_raw=cm.projmatrix_raw.to(device), sh_degree=sh_degree,
        campos=cm.campos.to(device), prefiltered=False, debug=False)
