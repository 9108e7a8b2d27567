"""create_viewpoint_pcd (HIP back-projection + device-side subset) against a plain PyTorch restatement of
/root/reference/gaussian_splatting/scene/gaussian_model.py:121-319 with the same selected subset.
The reference itself is not importable here (open3d): parity unpinned."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _frame(seed, H=96, W=128):
    g = torch.Generator().manual_seed(seed)
    from monogs_amd import camera as cam
    Tcw = cam.se3_exp(torch.tensor([0.1, -0.2, 0.3, 0.05, 0.02, -0.04]))
    depth = torch.rand(H, W, generator=g) * 4 + 0.5
    depth[torch.rand(H, W, generator=g) < 0.1] = 0.0                       # invalid depth
    vp = types.SimpleNamespace(
        rgb=torch.rand(3, H, W, generator=g).to(DEV), depth=depth.to(DEV),
        segmentation=torch.randint(0, 7, (H, W), generator=g).to(DEV),
        R=Tcw[:3, :3].contiguous().to(DEV), T=Tcw[:3, 3].contiguous().to(DEV),
        exposure_a=torch.tensor([0.3], device=DEV), exposure_b=torch.tensor([-0.1], device=DEV))
    intr = types.SimpleNamespace(fx=110.0, fy=105.0, cx=63.2, cy=47.9)
    rdepth = (depth + torch.randn(H, W, generator=g) * 0.05 + (torch.rand(H, W, generator=g) < 0.02) * 3.0).to(DEV)[None]
    ropac = torch.rand(1, H, W, generator=g).to(DEV)
    return vp, intr, rdepth, ropac


def _restatement(vp, intr, rdepth, ropac, init, pick):
    """Plain PyTorch, following the reference's steps: (W, H) ordering, mask, subset, K^-1, camera->world, scales."""
    from monogs_amd.knn import distCUDA2
    H, W = vp.depth.shape
    rgb = vp.rgb if init else (torch.exp(vp.exposure_a) * vp.rgb + vp.exposure_b).clamp(0, 1)
    col = rgb.permute(2, 1, 0).reshape(-1, 3)                 # (W, H, 3) flattened
    d = vp.depth.t().reshape(-1)
    ids = vp.segmentation.t().reshape(-1)
    mask = d >= 1e-3
    if not init:
        rd, ro = rdepth[0].t().reshape(-1), ropac[0].t().reshape(-1)
        err = (d - rd).abs()
        mask = mask & ((ro < 0.5) | ((d < rd) & (err > 50 * err.median())))
    xs, ys = torch.meshgrid(torch.arange(W, device=DEV), torch.arange(H, device=DEV), indexing="ij")
    uv = torch.stack([xs, ys], -1).float().reshape(-1, 2) + 0.5
    uv, d, col, ids = uv[mask], d[mask], col[mask], ids[mask]
    n = uv.shape[0]
    keep = int(n * (1.0 / (32 if init else 64)))
    pick = pick[:keep]
    uv, d, col, ids = uv[pick], d[pick], col[pick], ids[pick]
    K = torch.tensor([[intr.fx, 0, intr.cx], [0, intr.fy, intr.cy], [0, 0, 1]], device=DEV)
    pc = (torch.linalg.inv(K) @ torch.cat([uv, torch.ones_like(uv[:, :1])], 1)[..., None]).squeeze(-1) * d[:, None]
    w2c = torch.eye(4, device=DEV)
    w2c[:3, :3], w2c[:3, 3] = vp.R, vp.T
    c2w = torch.linalg.inv(w2c)
    pw = (c2w[:3, :3] @ pc.T).T + c2w[:3, 3][None]
    ps = torch.clamp_max(0.01 * vp.depth.median(), 0.05)
    scales = torch.log(torch.sqrt(torch.clamp_min(distCUDA2(pw.contiguous()), 1e-7) * ps))[:, None]
    return pw, col, scales, ids, int(mask.sum())


@pytest.mark.parametrize("init", [True, False])
def test_create_viewpoint_pcd_matches_restatement(native_lib, init):
    from monogs_amd.keyframe import create_viewpoint_pcd, densification_mask
    vp, intr, rdepth, ropac = _frame(3 if init else 4)
    n = int(densification_mask(vp.depth, None if init else rdepth, None if init else ropac, init).sum())
    pick = torch.randperm(n, generator=torch.Generator().manual_seed(9))        # what the reference draws on the CPU
    pts, feat, scales, rots, opac, ids = create_viewpoint_pcd(vp, intr, None if init else rdepth, None if init else ropac,
                                                              init=init, random_indices=pick)
    rp, rc, rs, rid, n_ref = _restatement(vp, intr, rdepth, ropac, init, pick.to(DEV))
    assert n == n_ref and pts.shape[0] == int(n * (1.0 / (32 if init else 64))) and pts.shape[0] > 10
    assert torch.allclose(pts, rp, rtol=1e-5, atol=2e-6)
    assert torch.allclose(feat, rc, atol=1e-6)
    assert torch.equal(ids.long(), rid.long())
    assert torch.allclose(scales, rs, rtol=1e-4, atol=1e-5)
    assert torch.equal(rots, torch.tensor([[1.0, 0, 0, 0]], device=DEV).expand_as(rots)) and float(opac.abs().max()) == 0.0


def test_create_viewpoint_pcd_device_subset(native_lib):
    """Without random_indices the subset is drawn on the device: right size, no duplicates, only masked pixels."""
    from monogs_amd.keyframe import create_viewpoint_pcd, densification_mask
    vp, intr, rdepth, ropac = _frame(5, H=120, W=160)
    g = torch.Generator(device=DEV).manual_seed(1)
    pts, feat, scales, rots, opac, ids = create_viewpoint_pcd(vp, intr, rdepth, ropac, init=False, generator=g)
    n = int(densification_mask(vp.depth, rdepth, ropac, False).sum())
    assert pts.shape[0] == n // 64 and torch.isfinite(pts).all() and torch.isfinite(scales).all()
    assert torch.unique(pts, dim=0).shape[0] == pts.shape[0]
    g2 = torch.Generator(device=DEV).manual_seed(1)
    again = create_viewpoint_pcd(vp, intr, rdepth, ropac, init=False, generator=g2)[0]
    assert torch.equal(pts, again)                                  # deterministic given the generator


def test_knn_against_the_whole_map_is_opt_in_and_exact(native_lib):
    """`knn_against=map_xyz` (the reference's own TODO, gaussian_model.py:293): the scale of a new Gaussian comes from its
    3 nearest neighbours among new AND existing points -- checked against the cKDTree oracle on the concatenated cloud;
    without the argument the result is the reference's (neighbours among the new points only)."""
    from monogs_amd.keyframe import create_viewpoint_pcd
    from oracle import dist2_knn
    vp, intr, rdepth, ropac = _frame(3)
    g = torch.Generator().manual_seed(9)
    pick = torch.randperm(int((vp.depth >= 1e-3).sum()), generator=g)       # a permutation of the CANDIDATE pixels
    ref = create_viewpoint_pcd(vp, intr, init=True, random_indices=pick)
    pts = ref[0]
    # an existing map: points scattered right around the new ones (so that the nearest neighbours change) + far clutter
    near = pts[::3] + 0.002 * torch.randn(pts[::3].shape, generator=g).to(DEV)
    far = (torch.rand(5000, 3, generator=g) * 40 - 20).to(DEV)
    map_xyz = torch.cat([near, far], 0)
    out = create_viewpoint_pcd(vp, intr, init=True, random_indices=pick, knn_against=map_xyz)
    assert torch.equal(out[0], pts) and torch.equal(out[1], ref[1])          # positions and colours unchanged
    ps = torch.clamp_max(0.01 * vp.depth.median(), 0.05).cpu()
    want = torch.log(torch.sqrt(torch.clamp_min(dist2_knn(torch.cat([pts, map_xyz], 0).cpu())[:pts.shape[0]], 1e-7) * ps))
    assert torch.allclose(out[2][:, 0].cpu(), want, rtol=1e-5, atol=1e-6)
    assert (out[2] < ref[2] - 1e-3).float().mean() > 0.2                     # a good share of the scales shrank
    assert not (out[2] > ref[2] + 1e-5).any()                               # more candidates can only bring neighbours closer
