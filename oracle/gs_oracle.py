"""PyTorch-CPU restatement of the tile-based differentiable Gaussian rasteriser with camera-pose
Jacobians that MonoGS calls through ``diff_gaussian_rasterization``.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py: parity unpinned; never imported by the product path).

What it follows
---------------
* boundary / tensor conventions:  /root/reference/gaussian_splatting/gaussian_renderer/__init__.py:26-168
* camera matrices (all stored TRANSPOSED, i.e. row-vector convention):
  /root/reference/utils/camera_utils.py:39-49,171-178,224-231 and
  /root/reference/gaussian_splatting/utils/graphics_utils.py:33-42,68-89
* pose retraction  T_cw <- exp([rho;theta]^) . T_cw :  /root/reference/utils/pose_utils.py:25-93
* per-splat maths (the only in-tree statement of it):
  /root/reference/viewer/gl_render/shaders/gau_vert.glsl:60-107,149-154,173-210 and gau_frag.glsl:20-25
* quaternion (r,x,y,z) -> R :  /root/reference/gaussian_splatting/utils/general_utils.py:113-136
* SH basis :  /root/reference/gaussian_splatting/utils/sh_utils.py:24-118
* everything else (near cull 0.2, +0.3 low-pass, 1.3 FoV clamp, radius, rect, key order, the
  1/255, 0.99, 1e-4 and 0.5 thresholds, depth = sum z.alpha.T, opacity = 1-T, n_touched) is the
  published algorithm of the un-vendored rasteriser as recorded in SURVEY.md Appendix A.

Two properties matter:

1. *Kernel-order float32 geometry.*  ``preprocess`` is written with one IEEE operation per
   torch op and a fixed association order (no fused multiply-add on CPU), which is the order
   the HIP preprocess kernel uses under ``-ffp-contract=off``.  In float32 the radii, tile
   rectangles, depth bits, sort keys and tile ranges are therefore comparable BIT-EXACTLY.
2. *Gradients from autograd.*  The pose dependence is made explicit
   (``T_cw(tau) = se3_exp(tau) @ T_cw``), so plain autograd yields dL/dtheta, dL/drho; the
   analytic backward of the HIP kernels is never restated here.  Three places deliberately use
   the reference's gradient convention instead of the literal derivative (documented inline):
   alpha's 0.99 clamp is straight-through, the 1.3.tanfov clamp freezes t.x/t.y, and skip /
   termination decisions are constants.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import numpy as np
import torch

BLOCK_X = 16
BLOCK_Y = 16

# /root/reference/gaussian_splatting/utils/sh_utils.py:24-52
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


class OracleSettings(NamedTuple):
    """Same 13 fields, same order, as ``GaussianRasterizationSettings`` (constructed at
    /root/reference/gaussian_splatting/gaussian_renderer/__init__.py:70-84)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# --------------------------------------------------------------------------------------------
# SE(3) retraction   (/root/reference/utils/pose_utils.py:13-73)
# --------------------------------------------------------------------------------------------
def _skew(x: torch.Tensor) -> torch.Tensor:
    z = torch.zeros((), dtype=x.dtype)
    return torch.stack([
        torch.stack([z, -x[2], x[1]]),
        torch.stack([x[2], z, -x[0]]),
        torch.stack([-x[1], x[0], z]),
    ])


def so3_exp(theta: torch.Tensor) -> torch.Tensor:
    W = _skew(theta)
    W2 = W @ W
    angle = torch.norm(theta)
    eye = torch.eye(3, dtype=theta.dtype)
    if float(angle.detach()) < 1e-5:
        return eye + W + 0.5 * W2
    return eye + (torch.sin(angle) / angle) * W + ((1 - torch.cos(angle)) / angle**2) * W2


def _V(theta: torch.Tensor) -> torch.Tensor:
    W = _skew(theta)
    W2 = W @ W
    angle = torch.norm(theta)
    eye = torch.eye(3, dtype=theta.dtype)
    if float(angle.detach()) < 1e-5:
        return eye + 0.5 * W + (1.0 / 6.0) * W2
    return eye + W * ((1.0 - torch.cos(angle)) / angle**2) + W2 * ((angle - torch.sin(angle)) / angle**3)


def se3_exp(tau: torch.Tensor) -> torch.Tensor:
    """tau = [rho (translation); theta (rotation)]  (/root/reference/utils/pose_utils.py:61-73)."""
    rho, theta = tau[:3], tau[3:]
    R = so3_exp(theta)
    t = _V(theta) @ rho
    top = torch.cat([R, t[:, None]], dim=1)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=tau.dtype)
    return torch.cat([top, bottom], dim=0)


def _posed_matrices(settings, theta, rho, dtype):
    """Return (viewmatrix, projmatrix, campos) whose VALUES are exactly the caller's tensors and
    whose DERIVATIVE w.r.t. (rho, theta) is that of  T_cw(tau) = se3_exp(tau) @ T_cw  at the
    caller's tau (MonoGS always calls with tau = 0: /root/reference/utils/pose_utils.py:91-92)."""
    V = settings.viewmatrix.detach().to(dtype)
    PM = settings.projmatrix.detach().to(dtype)
    campos = settings.campos.detach().to(dtype)
    if theta is None and rho is None:
        return V, PM, campos
    zero3 = torch.zeros(3, dtype=dtype)
    rho_ = rho.to(dtype) if rho is not None else zero3
    theta_ = theta.to(dtype) if theta is not None else zero3
    tau = torch.cat([rho_, theta_])
    T_cw = V.t()
    P = settings.projmatrix_raw.detach().to(dtype).t()
    T_new = se3_exp(tau) @ T_cw
    V_new = T_new.t()
    PM_new = (P @ T_new).t()
    campos_new = -(T_new[:3, :3].t() @ T_new[:3, 3])
    V = V + (V_new - V_new.detach())
    PM = PM + (PM_new - PM_new.detach())
    campos = campos + (campos_new - campos_new.detach())
    return V, PM, campos


# --------------------------------------------------------------------------------------------
# per-Gaussian preprocess  (K1)
# --------------------------------------------------------------------------------------------
def _dot3p(m0, m1, m2, m3, x, y, z):
    """((m0*x + m1*y) + m2*z) + m3 -- the association the HIP kernel uses."""
    return ((m0 * x + m1 * y) + m2 * z) + m3


def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """(r,x,y,z), NOT normalised here (the model normalises before the call:
    /root/reference/gaussian_splatting/scene/gaussian_model.py:89-90); layout of
    /root/reference/gaussian_splatting/utils/general_utils.py:113-136."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y),
    ], dim=1).reshape(-1, 3, 3)
    return R


def cov3d_from_scale_rot(scales, rotations, mod):
    """Sigma = R S^2 R^T as 6 upper-triangular floats (xx,xy,xz,yy,yz,zz)
    (/root/reference/viewer/gl_render/shaders/gau_vert.glsl:60-80)."""
    R = quat_to_rotmat(rotations)
    s = scales * mod
    M = R * s[:, None, :]                       # M_ij = R_ij * s_j
    def e(i, j):
        return (M[:, i, 0] * M[:, j, 0] + M[:, i, 1] * M[:, j, 1]) + M[:, i, 2] * M[:, j, 2]
    return torch.stack([e(0, 0), e(0, 1), e(0, 2), e(1, 1), e(1, 2), e(2, 2)], dim=1)


def eval_sh_color(deg, sh, dirs):
    """sh: [P, M, 3]; dirs: [P,3] unit.  Same polynomial as
    /root/reference/gaussian_splatting/utils/sh_utils.py:55-118 (deg 0..3), then +0.5 and
    max(0, .) as the rasteriser does."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9]
                       + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    res = res + 0.5
    return torch.clamp_min(res, 0.0)


def preprocess(means3D, scales, rotations, opacities, settings, *, means2D=None, shs=None,
               colors_precomp=None, cov3D_precomp=None, theta=None, rho=None, dtype=None):
    """Per-Gaussian projection.  Returns a dict of per-Gaussian tensors; the float ones are
    differentiable, the integer ones (radii, rects, tiles_touched) are decisions."""
    dtype = dtype or means3D.dtype
    P = means3D.shape[0]
    H, W = int(settings.image_height), int(settings.image_width)
    c = lambda v: torch.tensor(v, dtype=dtype)  # noqa: E731
    tanfovx = c(float(np.float32(settings.tanfovx)) if dtype == torch.float32 else settings.tanfovx)
    tanfovy = c(float(np.float32(settings.tanfovy)) if dtype == torch.float32 else settings.tanfovy)
    focal_x = c(W) / (c(2.0) * tanfovx)
    focal_y = c(H) / (c(2.0) * tanfovy)
    V, PM, campos = _posed_matrices(settings, theta, rho, dtype)
    Vf, PMf = V.reshape(-1), PM.reshape(-1)      # flat index 4*j+i = maths element (i,j)

    m = means3D.to(dtype)
    x, y, z = m[:, 0], m[:, 1], m[:, 2]
    pv = [_dot3p(Vf[k], Vf[4 + k], Vf[8 + k], Vf[12 + k], x, y, z) for k in range(3)]
    ph = [_dot3p(PMf[k], PMf[4 + k], PMf[8 + k], PMf[12 + k], x, y, z) for k in range(4)]
    p_w = c(1.0) / (ph[3] + c(0.0000001))
    proj_x, proj_y = ph[0] * p_w, ph[1] * p_w
    if means2D is not None:                       # gradient holder, value 0
        proj_x = proj_x + means2D[:, 0].to(dtype)
        proj_y = proj_y + means2D[:, 1].to(dtype)
    depth = pv[2]
    in_front = depth > c(0.2)

    if cov3D_precomp is not None:
        cov3D = cov3D_precomp.to(dtype)
    else:
        mod = c(float(np.float32(settings.scale_modifier)) if dtype == torch.float32
                else float(settings.scale_modifier))
        cov3D = cov3d_from_scale_rot(scales.to(dtype), rotations.to(dtype), mod)

    # ---- 2-D covariance (EWA), /root/reference/viewer/gl_render/shaders/gau_vert.glsl:82-107
    limx, limy = c(1.3) * tanfovx, c(1.3) * tanfovy
    tz = pv[2]
    txtz, tytz = pv[0] / tz, pv[1] / tz
    cx = torch.minimum(limx, torch.maximum(-limx, txtz))
    cy = torch.minimum(limy, torch.maximum(-limy, tytz))
    tx_raw, ty_raw = cx * tz, cy * tz
    # reference convention: a clamped t.x / t.y is a constant of the backward pass
    tx = torch.where((txtz < -limx) | (txtz > limx), tx_raw.detach(), tx_raw)
    ty = torch.where((tytz < -limy) | (tytz > limy), ty_raw.detach(), ty_raw)
    tz2 = tz * tz
    J00 = focal_x / tz
    J02 = -(focal_x * tx) / tz2
    J11 = focal_y / tz
    J12 = -(focal_y * ty) / tz2
    Rv = lambda i, j: Vf[4 * j + i]               # noqa: E731  rotation block of T_cw
    T0 = [J00 * Rv(0, j) + J02 * Rv(2, j) for j in range(3)]
    T1 = [J11 * Rv(1, j) + J12 * Rv(2, j) for j in range(3)]
    S = [[cov3D[:, 0], cov3D[:, 1], cov3D[:, 2]],
         [cov3D[:, 1], cov3D[:, 3], cov3D[:, 4]],
         [cov3D[:, 2], cov3D[:, 4], cov3D[:, 5]]]
    U0 = [(T0[0] * S[0][j] + T0[1] * S[1][j]) + T0[2] * S[2][j] for j in range(3)]
    U1 = [(T1[0] * S[0][j] + T1[1] * S[1][j]) + T1[2] * S[2][j] for j in range(3)]
    cov_xx = ((U0[0] * T0[0] + U0[1] * T0[1]) + U0[2] * T0[2]) + c(0.3)
    cov_xy = (U0[0] * T1[0] + U0[1] * T1[1]) + U0[2] * T1[2]
    cov_yy = ((U1[0] * T1[0] + U1[1] * T1[1]) + U1[2] * T1[2]) + c(0.3)
    det = cov_xx * cov_yy - cov_xy * cov_xy
    det_ok = det != 0
    det_safe = torch.where(det_ok, det, torch.ones_like(det))
    det_inv = c(1.0) / det_safe
    conic = torch.stack([cov_yy * det_inv, -cov_xy * det_inv, cov_xx * det_inv], dim=1)

    with torch.no_grad():
        mid = c(0.5) * (cov_xx + cov_yy)
        disc = torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        lam = torch.maximum(mid + disc, mid - disc)
        radius_f = torch.ceil(c(3.0) * torch.sqrt(lam))
    px = ((proj_x + c(1.0)) * c(W) - c(1.0)) * c(0.5)
    py = ((proj_y + c(1.0)) * c(H) - c(1.0)) * c(0.5)

    grid_x = (W + BLOCK_X - 1) // BLOCK_X
    grid_y = (H + BLOCK_Y - 1) // BLOCK_Y
    with torch.no_grad():
        ok = in_front & det_ok
        ok = ok & torch.isfinite(px) & torch.isfinite(py) & torch.isfinite(radius_f)

        def _trunc_clamp(v, hi):
            v = torch.where(ok, v, torch.zeros_like(v))
            v = torch.clamp(torch.trunc(v), -2.0e9, 2.0e9).to(torch.int64)   # (int) cast
            return torch.clamp(v, 0, hi)
        rmin_x = _trunc_clamp((px - radius_f) / c(BLOCK_X), grid_x)
        rmin_y = _trunc_clamp((py - radius_f) / c(BLOCK_Y), grid_y)
        rmax_x = _trunc_clamp(((px + radius_f) + c(BLOCK_X - 1)) / c(BLOCK_X), grid_x)
        rmax_y = _trunc_clamp(((py + radius_f) + c(BLOCK_Y - 1)) / c(BLOCK_Y), grid_y)
        tiles = (rmax_x - rmin_x) * (rmax_y - rmin_y)
        visible = ok & (tiles > 0)
        tiles = torch.where(visible, tiles, torch.zeros_like(tiles))
        radii = torch.where(visible, radius_f, torch.zeros_like(radius_f)).to(torch.int32)

    if colors_precomp is not None:
        color = colors_precomp.to(dtype)
    else:
        d = m - campos[None, :]
        d = d / torch.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])[:, None]
        color = eval_sh_color(int(settings.sh_degree), shs.to(dtype), d)

    return dict(depth=depth, xy=torch.stack([px, py], dim=1), conic=conic,
                opacity=opacities.to(dtype).reshape(P), color=color, cov3D=cov3D,
                cov2D=torch.stack([cov_xx, cov_xy, cov_yy], dim=1),
                radii=radii, rect_min=torch.stack([rmin_x, rmin_y], 1),
                rect_max=torch.stack([rmax_x, rmax_y], 1), tiles_touched=tiles.to(torch.int64),
                visible=visible, grid=(grid_x, grid_y))


# --------------------------------------------------------------------------------------------
# binning  (K2-K5)
# --------------------------------------------------------------------------------------------
def build_binning(geom):
    """duplicate-with-keys + stable sort + tile ranges.  key = tile_id << 32 | float32 bits of
    the view-space depth; ties keep emission order (Gaussian index ascending), which is what a
    stable LSD radix sort of the emitted list gives.  Returns
    (point_list int64[R], ranges int64[tiles,2], keys uint64[R] sorted, offsets int64[P])."""
    grid_x, grid_y = geom["grid"]
    tiles = geom["tiles_touched"].numpy()
    offsets = np.cumsum(tiles)
    R = int(offsets[-1]) if len(offsets) else 0
    depth_bits = geom["depth"].detach().to(torch.float32).numpy().view(np.uint32).astype(np.uint64)
    rmin = geom["rect_min"].numpy()
    rmax = geom["rect_max"].numpy()
    vis = np.nonzero(tiles > 0)[0]
    keys = np.empty(R, dtype=np.uint64)
    vals = np.empty(R, dtype=np.int64)
    w = (rmax[vis, 0] - rmin[vis, 0]).astype(np.int64)
    cnt = tiles[vis].astype(np.int64)
    starts = offsets[vis] - cnt
    # vectorised emission in (y outer, x inner) order per Gaussian
    rep = np.repeat(np.arange(len(vis)), cnt)
    local = np.arange(R, dtype=np.int64) - np.repeat(starts, cnt)
    ty = rmin[vis, 1][rep] + local // w[rep]
    tx = rmin[vis, 0][rep] + local % w[rep]
    tile_id = (ty * grid_x + tx).astype(np.uint64)
    keys[:] = (tile_id << np.uint64(32)) | depth_bits[vis][rep]
    vals[:] = vis[rep]
    order = np.argsort(keys, kind="stable")
    keys_s = keys[order]
    vals_s = vals[order]
    ntiles = grid_x * grid_y
    tile_s = (keys_s >> np.uint64(32)).astype(np.int64)
    ranges = np.zeros((ntiles, 2), dtype=np.int64)
    if R:
        starts_t = np.searchsorted(tile_s, np.arange(ntiles), side="left")
        ends_t = np.searchsorted(tile_s, np.arange(ntiles), side="right")
        nonempty = ends_t > starts_t
        ranges[nonempty, 0] = starts_t[nonempty]
        ranges[nonempty, 1] = ends_t[nonempty]
    return torch.from_numpy(vals_s), torch.from_numpy(ranges), keys_s, torch.from_numpy(offsets)


# --------------------------------------------------------------------------------------------
# alpha blending  (K6) -- differentiable
# --------------------------------------------------------------------------------------------
class OracleOutput(NamedTuple):
    color: torch.Tensor       # [3,H,W]
    radii: torch.Tensor       # [P] int32
    depth: torch.Tensor       # [1,H,W]
    opacity: torch.Tensor     # [1,H,W]
    n_touched: torch.Tensor   # [P] int32
    aux: dict


def _blend_tiles(pix, inside, ids, pad, geom, bg, n_touched, want_ambiguous):
    """A batch of B tiles at once.  pix: [B,256,2] pixel coords; inside: [B,256]; ids: [B,n] instances of each tile in
    blend order, padded at the END with 0 where ``pad`` [B,n] is True (a padded slot is skipped by every pixel, so it
    changes nothing: T, the sums and n_contrib ignore it).  Follows the per-pixel loop of SURVEY.md section 2.1 K6 /
    Appendix A.  Tiles are batched only to amortise the interpreter: each row of the batch is an independent tile."""
    dtype = pix.dtype
    xy = geom["xy"][ids]               # [B,n,2]
    con = geom["conic"][ids]
    op = geom["opacity"][ids]
    col = geom["color"][ids]
    z = geom["depth"][ids]
    dx = xy[:, None, :, 0] - pix[:, :, None, 0]          # [B,256,n]
    dy = xy[:, None, :, 1] - pix[:, :, None, 1]
    power = -0.5 * (con[:, None, :, 0] * dx * dx + con[:, None, :, 2] * dy * dy) - con[:, None, :, 1] * dx * dy
    a_raw = op[:, None, :] * torch.exp(power)
    # reference convention: the 0.99 clamp is ignored by the backward pass (straight-through)
    alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()
    with torch.no_grad():
        skip = (power > 0) | (alpha < 1.0 / 255.0) | (~inside[:, :, None]) | pad[:, None, :]
    a_eff = torch.where(skip, torch.zeros_like(alpha), alpha)
    one_m = 1.0 - a_eff
    T_incl = torch.cumprod(one_m, dim=2)
    T_before = torch.cat([torch.ones_like(T_incl[:, :, :1]), T_incl[:, :, :-1]], dim=2)
    amb = None
    with torch.no_grad():
        test_T = T_before * one_m
        stop = (~skip) & (test_T < 0.0001)
        stopped = torch.cumsum(stop.to(torch.int32), dim=2) > 0       # at and after the stop
        valid = (~skip) & (~stopped)
        n = ids.shape[1]
        idx1 = torch.arange(1, n + 1)[None, None, :].expand_as(valid)
        n_contrib = torch.where(valid, idx1, torch.zeros_like(idx1)).amax(dim=2)
        touched = (valid & (test_T > 0.5)).sum(dim=1)                 # [B,n]
        n_touched.index_add_(0, ids.reshape(-1), torch.where(pad, torch.zeros_like(touched), touched)
                             .reshape(-1).to(n_touched.dtype))
        if want_ambiguous:
            # decisions within a few float32 ulps of their threshold: a different (equally
            # valid) exp / rounding may flip them, so parity tests may exempt these pixels
            near = lambda v, t, r: (v - t).abs() <= r * abs(t)          # noqa: E731
            a = (~stopped) & inside[:, :, None] & (~pad[:, None, :]) & (
                near(alpha.detach(), 1.0 / 255.0, 2e-5) | ((~skip) & near(test_T, 0.0001, 2e-5))
                | ((~skip) & near(test_T, 0.5, 2e-6)))
            amb = a.any(dim=2)
    w = torch.where(valid, a_eff * T_before, torch.zeros_like(a_eff))
    C = torch.bmm(w, col)                                     # [B,256,3]
    D = torch.bmm(w, z[:, :, None])[:, :, 0]
    T_final = torch.prod(torch.where(valid, one_m, torch.ones_like(one_m)), dim=2)
    out_c = C + T_final[:, :, None] * bg[None, None, :]
    return out_c, D, T_final, n_contrib.to(dtype), amb


# tiles per batch: bounded by the size of one [B,256,n] temporary (elements) and by padding waste
_BATCH_ELEMS = 1 << 21
_BATCH_PAD = 1.25


def rasterize(means3D, means2D, opacities, settings, *, shs=None, colors_precomp=None,
              scales=None, rotations=None, cov3D_precomp=None, theta=None, rho=None,
              dtype=None, want_ambiguous=False) -> OracleOutput:
    """Same argument meaning as ``GaussianRasterizer.forward`` (called at
    /root/reference/gaussian_splatting/gaussian_renderer/__init__.py:145-156).
    Returns (color[3,H,W], radii[P], depth[1,H,W], opacity[1,H,W], n_touched[P], aux)."""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    dtype = dtype or means3D.dtype
    H, W = int(settings.image_height), int(settings.image_width)
    P = means3D.shape[0]
    geom = preprocess(means3D, scales, rotations, opacities, settings, means2D=means2D, shs=shs,
                      colors_precomp=colors_precomp, cov3D_precomp=cov3D_precomp,
                      theta=theta, rho=rho, dtype=dtype)
    point_list, ranges, keys, offsets = build_binning(geom)
    grid_x, grid_y = geom["grid"]
    ntiles = grid_x * grid_y
    bg = settings.bg.detach().to(dtype)
    n_touched = torch.zeros(P, dtype=torch.int64)
    # default (empty tile): colour = bg, depth 0, T = 1
    lx = torch.arange(BLOCK_X).repeat(BLOCK_Y)
    ly = torch.arange(BLOCK_Y).repeat_interleave(BLOCK_X)
    out_c = [None] * ntiles
    out_d = [None] * ntiles
    out_T = [None] * ntiles
    out_n = [None] * ntiles
    amb_tiles = [None] * ntiles
    empty_c = bg[None, :].expand(BLOCK_X * BLOCK_Y, 3)
    empty_d = torch.zeros(BLOCK_X * BLOCK_Y, dtype=dtype)
    empty_T = torch.ones(BLOCK_X * BLOCK_Y, dtype=dtype)
    empty_b = torch.zeros(BLOCK_X * BLOCK_Y, dtype=torch.bool)
    lens = (ranges[:, 1] - ranges[:, 0]).tolist()
    for t in range(ntiles):
        if lens[t] <= 0:
            out_c[t], out_d[t], out_T[t], out_n[t] = empty_c, empty_d, empty_T, empty_d
            amb_tiles[t] = empty_b
    # non-empty tiles, shortest first, in batches of similar list length
    order = sorted((t for t in range(ntiles) if lens[t] > 0), key=lambda t: lens[t])
    i = 0
    while i < len(order):
        n0 = lens[order[i]]
        j = i + 1
        while j < len(order) and lens[order[j]] <= max(n0 * _BATCH_PAD, n0 + 8) \
                and (j + 1 - i) * 256 * lens[order[j]] <= _BATCH_ELEMS:
            j += 1
        batch = order[i:j]
        i = j
        n = lens[batch[-1]]
        B = len(batch)
        ids = torch.zeros(B, n, dtype=torch.int64)
        pad = torch.ones(B, n, dtype=torch.bool)
        for r, t in enumerate(batch):
            s, e = int(ranges[t, 0]), int(ranges[t, 1])
            ids[r, :e - s] = point_list[s:e]
            pad[r, :e - s] = False
        tt = torch.tensor(batch, dtype=torch.int64)
        pxi = (tt % grid_x)[:, None] * BLOCK_X + lx[None, :]
        pyi = (tt // grid_x)[:, None] * BLOCK_Y + ly[None, :]
        inside = (pxi < W) & (pyi < H)
        pix = torch.stack([pxi, pyi], dim=2).to(dtype)
        c, d, T, nc, amb = _blend_tiles(pix, inside, ids, pad, geom, bg, n_touched, want_ambiguous)
        for r, t in enumerate(batch):
            out_c[t], out_d[t], out_T[t], out_n[t] = c[r], d[r], T[r], nc[r]
            if want_ambiguous:
                amb_tiles[t] = amb[r]

    def _stitch(parts, ch):
        a = torch.stack(parts, dim=0).reshape(grid_y, grid_x, BLOCK_Y, BLOCK_X, ch)
        a = a.permute(4, 0, 2, 1, 3).reshape(ch, grid_y * BLOCK_Y, grid_x * BLOCK_X)
        return a[:, :H, :W]
    color = _stitch(out_c, 3)
    depth = _stitch([d[:, None] for d in out_d], 1)
    final_T = _stitch([d[:, None] for d in out_T], 1)
    n_contrib = _stitch([d[:, None] for d in out_n], 1)
    opacity = 1.0 - final_T
    aux = dict(geom=geom, point_list=point_list, ranges=ranges, keys=keys, offsets=offsets,
               final_T=final_T.detach(), n_contrib=n_contrib.detach().to(torch.int64),
               num_rendered=int(point_list.shape[0]))
    if want_ambiguous:
        aux["ambiguous"] = _stitch([a[:, None].to(dtype) for a in amb_tiles], 1)[0] > 0
    return OracleOutput(color, geom["radii"], depth, opacity, n_touched.to(torch.int32), aux)


def rasterize_autograd(inputs: dict, settings, grad_color, grad_depth, dtype=torch.float64,
                       want_ambiguous=False):
    """Run the oracle forward and pull (grad_color, grad_depth) back with autograd.
    ``inputs``: means3D, opacities, and (colors_precomp | shs), (scales, rotations | cov3D_precomp).
    Returns (OracleOutput, grads dict) with grads for every tensor input plus means2D, theta, rho.
    The opacity image receives no gradient (grad_opacity is ignored)."""
    leaves = {}
    for k, v in inputs.items():
        if v is None:
            continue
        leaves[k] = v.detach().to(dtype).clone().requires_grad_(True)
    P = leaves["means3D"].shape[0]
    leaves["means2D"] = torch.zeros(P, 3, dtype=dtype, requires_grad=True)
    leaves["theta"] = torch.zeros(3, dtype=dtype, requires_grad=True)
    leaves["rho"] = torch.zeros(3, dtype=dtype, requires_grad=True)
    out = rasterize(leaves["means3D"], leaves["means2D"], leaves["opacities"], settings,
                    shs=leaves.get("shs"), colors_precomp=leaves.get("colors_precomp"),
                    scales=leaves.get("scales"), rotations=leaves.get("rotations"),
                    cov3D_precomp=leaves.get("cov3D_precomp"), theta=leaves["theta"],
                    rho=leaves["rho"], dtype=dtype, want_ambiguous=want_ambiguous)
    loss = (out.color * grad_color.to(dtype)).sum() + (out.depth * grad_depth.to(dtype)).sum()
    names = list(leaves.keys())
    gs = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in zip(names, gs)}
    return out, grads
