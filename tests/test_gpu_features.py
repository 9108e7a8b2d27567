"""HIP path vs oracle for the rest of the API surface and the edge cases (pytest -m gpu)."""
import types

import pytest
import torch

from monogs_amd import camera as cam
from monogs_amd.synthetic import make_scene, scene_settings
from oracle import OracleSettings, gs_oracle, rasterize, rasterize_autograd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hip_st(sc, **kw):
    from monogs_amd.rasterizer import GaussianRasterizationSettings
    return scene_settings(sc, GaussianRasterizationSettings, device=DEV, **kw)


def _run(sc, inp, st, ost, rtol=1e-3):
    from monogs_amd.rasterizer import GaussianRasterizer
    leaves = {k: v.to(DEV).clone().requires_grad_(True) for k, v in inp.items()}
    m2d = torch.zeros_like(leaves["means3D"], requires_grad=True)
    theta = torch.zeros(3, device=DEV, requires_grad=True)
    rho = torch.zeros(3, device=DEV, requires_grad=True)
    out = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2d, opacities=leaves["opacities"],
                                 shs=leaves.get("shs"), colors_precomp=leaves.get("colors_precomp"),
                                 scales=leaves.get("scales"), rotations=leaves.get("rotations"),
                                 cov3D_precomp=leaves.get("cov3D_precomp"), theta=theta, rho=rho)
    (out[0] * sc.grad_color.to(DEV)).sum().add((out[2] * sc.grad_depth.to(DEV)).sum()).backward()
    grads = {k: v.grad.cpu() for k, v in leaves.items()}
    grads.update(means2D=m2d.grad.cpu(), theta=theta.grad.cpu(), rho=rho.grad.cpu())
    oout, og = rasterize_autograd(inp, ost, sc.grad_color, sc.grad_depth, dtype=torch.float32, want_ambiguous=True)
    ok = ~oout.aux["ambiguous"]
    assert torch.equal(out[1].cpu(), oout.radii)
    assert (out[0].cpu() - oout.color).abs().amax(0)[ok].max() <= 1e-4
    assert (out[2].cpu() - oout.depth).abs()[0][ok].max() <= 1e-4 * max(1.0, oout.depth.max().item())
    assert (out[3].cpu() - oout.opacity).abs()[0][ok].max() <= 1e-4
    for k, ref in og.items():
        got = grads[k].reshape(ref.shape).double()
        ref = ref.double()
        if ref.abs().max() == 0:
            assert got.abs().max() == 0, k
            continue
        rel = ((got - ref).norm() / ref.norm()).item()
        assert rel <= rtol, f"{k}: rel L2 {rel:.2e}"
    return out, grads, oout, og


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colours(native_lib, deg):
    sc = make_scene(3000, "fr3_office", seed=20 + deg, anisotropic=True)
    g = torch.Generator().manual_seed(deg)
    shs = 0.4 * torch.randn(3000, 16, 3, generator=g)
    shs[:, 0] += 0.5
    inp = dict(means3D=sc.means3D, opacities=sc.opacities, shs=shs, scales=sc.scales, rotations=sc.rotations)
    _, grads, _, og = _run(sc, inp, _hip_st(sc, sh_degree=deg), scene_settings(sc, OracleSettings, sh_degree=deg))
    # coefficients above the active degree receive exactly zero
    n_active = (deg + 1) ** 2
    if n_active < 16:
        assert grads["shs"][:, n_active:].abs().max() == 0


def test_cov3d_precomp_and_scale_modifier(native_lib):
    sc = make_scene(3000, "fr3_office", seed=31, anisotropic=True)
    cov = gs_oracle.cov3d_from_scale_rot(sc.scales, sc.rotations, torch.tensor(1.0))
    inp = dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=sc.colors, cov3D_precomp=cov)
    _run(sc, inp, _hip_st(sc), scene_settings(sc, OracleSettings))
    inp2 = dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=sc.colors, scales=sc.scales,
                rotations=sc.rotations)
    _run(sc, inp2, _hip_st(sc, scale_modifier=1.7), scene_settings(sc, OracleSettings, scale_modifier=1.7))


def test_fov_clamp_and_offscreen_gaussians(native_lib):
    """spread 1.6 puts Gaussians beyond the 1.3 tanFoV guard (clamped Jacobian, frozen in the backward)."""
    sc = make_scene(4000, "fr3_office", seed=41, spread=1.6, mean_radius_px=12.0)
    inp = dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=sc.colors,
               scales=sc.scales.repeat(1, 3), rotations=sc.rotations)
    _run(sc, inp, _hip_st(sc), scene_settings(sc, OracleSettings))


def test_huge_and_tiny_gaussians(native_lib):
    """A few Gaussians covering hundreds of tiles plus sub-pixel ones; odd image size (partial tiles)."""
    intr = dict(fx=300.0, fy=310.0, cx=161.3, cy=117.9, W=325, H=237)
    sc = make_scene(600, intr, seed=51)
    scales = sc.scales.repeat(1, 3).clone()
    scales[:5] *= 60.0
    scales[5:200] *= 0.02
    inp = dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=sc.colors, scales=scales,
               rotations=sc.rotations)
    out, *_ = _run(sc, inp, _hip_st(sc), scene_settings(sc, OracleSettings))
    assert out[0].shape == (3, 237, 325)


def test_dense_tile_many_instances(native_lib):
    """> 64 and > 4096 instances in one tile: multi-step walks and early termination."""
    intr = dict(fx=200.0, fy=200.0, cx=32.0, cy=32.0, W=64, H=64)
    sc = make_scene(30000, intr, seed=61, mean_radius_px=10.0)
    inp = dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=sc.colors,
               scales=sc.scales.repeat(1, 3), rotations=sc.rotations)
    _, _, oout, _ = _run(sc, inp, _hip_st(sc), scene_settings(sc, OracleSettings))
    r = oout.aux["ranges"]
    assert (r[:, 1] - r[:, 0]).max() > 4096


def test_empty_and_degenerate_inputs(native_lib):
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(50, "fr3_office", seed=71, bg=(0.25, 0.5, 0.75))
    st = _hip_st(sc)
    # P = 0: background only
    z = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
    out = GaussianRasterizer(st)(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), colors_precomp=z(0, 3),
                                 scales=z(0, 3), rotations=z(0, 4))
    assert torch.allclose(out[0], torch.tensor([0.25, 0.5, 0.75], device=DEV)[:, None, None].expand(3, 480, 640))
    assert out[3].abs().max() == 0 and out[1].numel() == 0
    # everything behind the camera: num_rendered = 0
    means = sc.means3D.clone().to(DEV)
    behind = (torch.tensor([0.0, 0.0, -5.0]) - sc.t) @ sc.R
    means[:] = behind.to(DEV)
    m = means.clone().requires_grad_(True)
    out = GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), opacities=sc.opacities.to(DEV),
                                 colors_precomp=sc.colors.to(DEV), scales=sc.scales.repeat(1, 3).to(DEV),
                                 rotations=sc.rotations.to(DEV))
    assert (out[1] == 0).all() and out[3].abs().max() == 0 and (out[4] == 0).all()
    out[0].sum().backward()
    assert m.grad.abs().max() == 0


def test_mark_visible(native_lib):
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(5000, "fr3_office", seed=81, near_fraction=0.2)
    st = _hip_st(sc)
    vis = GaussianRasterizer(st).markVisible(sc.means3D.to(DEV)).cpu()
    pv = sc.means3D @ sc.R.t() + sc.t
    want = pv[:, 2] > 0.2
    assert (vis != want).sum() <= 2        # float32 association differs from the matmul above


class _Intr:
    def __init__(self, k):
        self.k = k
        self.height, self.width = k["H"], k["W"]
        m = cam.camera_matrices(torch.eye(3), torch.zeros(3), k["fx"], k["fy"], k["cx"], k["cy"], k["W"], k["H"])
        self.projection_matrix = m.projmatrix_raw.to(DEV)
        import math
        self.FoVx, self.FoVy = 2 * math.atan(m.tanfovx), 2 * math.atan(m.tanfovy)


def test_render_seam_end_to_end(native_lib):
    """monogs_amd.renderer.render() with duck-typed camera objects: dict keys, isotropic scale expansion,
    viewspace_points.grad, pose-delta gradients, and the (fixed) mask branch."""
    from monogs_amd.renderer import render
    sc = make_scene(4000, "fr3_office", seed=91)
    intr = _Intr(sc.intr)
    view_t = cam.world2view(sc.R, sc.t).transpose(0, 1).contiguous().to(DEV)
    vp = types.SimpleNamespace(world_view_transform=view_t, camera_center=view_t.inverse()[3, :3],
                               cam_rot_delta=torch.zeros(3, device=DEV, requires_grad=True),
                               cam_trans_delta=torch.zeros(3, device=DEV, requires_grad=True))
    leaf = lambda t: t.to(DEV).clone().requires_grad_(True)  # noqa: E731
    means, rot, sca, opa, col = leaf(sc.means3D), leaf(sc.rotations), leaf(sc.scales), leaf(sc.opacities), leaf(sc.colors)
    pkg = render(vp, intr, means, rot, sca, opa, col, sc.bg.to(DEV))
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii", "depth", "opacity", "n_touched"}
    loss = (pkg["render"] * sc.grad_color.to(DEV)).sum() + (pkg["depth"] * sc.grad_depth.to(DEV)).sum()
    loss.backward()
    inp = dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=sc.colors,
               scales=sc.scales.repeat(1, 3), rotations=sc.rotations)
    oout, og = rasterize_autograd(inp, scene_settings(sc, OracleSettings), sc.grad_color, sc.grad_depth,
                                  dtype=torch.float32)
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()  # noqa: E731
    assert rel(pkg["viewspace_points"].grad.cpu(), og["means2D"]) < 1e-3
    assert rel(vp.cam_rot_delta.grad.cpu(), og["theta"]) < 1e-3
    assert rel(vp.cam_trans_delta.grad.cpu(), og["rho"]) < 1e-3
    assert rel(sca.grad.cpu(), og["scales"].sum(1, keepdim=True)) < 1e-3      # isotropic: grad of repeat(1,3)
    assert torch.equal(pkg["visibility_filter"].cpu(), oout.radii > 0)
    # mask branch
    mask = torch.zeros(4000, dtype=torch.bool, device=DEV)
    mask[::2] = True
    pkg2 = render(vp, intr, means, rot, sca, opa, col, sc.bg.to(DEV), mask=mask)
    assert pkg2["radii"].shape[0] == 2000 and pkg2["n_touched"].shape[0] == 2000


def test_multiple_forwards_before_one_backward(native_lib):
    """The mapper renders every window keyframe, sums the losses and calls backward once
    (/root/reference/utils/slam_mapper.py:273-394): per-call scratch must not be shared."""
    from monogs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    sc = make_scene(3000, "fr3_office", seed=101)
    leaf = lambda t: t.to(DEV).clone().requires_grad_(True)  # noqa: E731
    means, opa, col, rot = leaf(sc.means3D), leaf(sc.opacities), leaf(sc.colors), leaf(sc.rotations)
    scales = leaf(sc.scales.repeat(1, 3))
    total = 0
    ref = torch.zeros_like(sc.means3D)
    for i in range(3):
        d = cam.se3_exp(torch.tensor([0.03 * i, -0.02 * i, 0.0, 0.0, 0.01 * i, 0.0]))
        T = torch.eye(4)
        T[:3, :3], T[:3, 3] = sc.R, sc.t
        T = d @ T
        sci = sc._replace(R=T[:3, :3].contiguous(), t=T[:3, 3].contiguous())
        st = scene_settings(sci, GaussianRasterizationSettings, device=DEV)
        out = GaussianRasterizer(st)(means3D=means, means2D=torch.zeros_like(means), opacities=opa,
                                     colors_precomp=col, scales=scales, rotations=rot)
        total = total + (out[0] * sc.grad_color.to(DEV)).sum() + (out[2] * sc.grad_depth.to(DEV)).sum()
        _, og = rasterize_autograd(dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=sc.colors,
                                        scales=sc.scales.repeat(1, 3), rotations=sc.rotations),
                                   scene_settings(sci, OracleSettings), sc.grad_color, sc.grad_depth,
                                   dtype=torch.float32)
        ref += og["means3D"]
    total.backward()
    assert ((means.grad.cpu() - ref).norm() / ref.norm()).item() < 1e-3


def test_second_backward_through_one_forward(native_lib):
    """The forward prepares (clears) the scratch of ONE backward; a second backward through the same forward
    (retain_graph=True) must take the clearing path and give the same gradients, pose gradient included."""
    from monogs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    sc = make_scene(4000, "fr3_office", seed=77)
    leaf = lambda t: t.to(DEV).clone().requires_grad_(True)  # noqa: E731
    means, opa, col, rot, scales = leaf(sc.means3D), leaf(sc.opacities), leaf(sc.colors), leaf(sc.rotations), leaf(sc.scales)
    theta, rho = torch.zeros(3, device=DEV, requires_grad=True), torch.zeros(3, device=DEV, requires_grad=True)
    st = scene_settings(sc, GaussianRasterizationSettings, device=DEV)
    out = GaussianRasterizer(st)(means3D=means, means2D=torch.zeros_like(means), opacities=opa, colors_precomp=col,
                                 scales=scales, rotations=rot, theta=theta, rho=rho)
    loss = (out[0] * sc.grad_color.to(DEV)).sum() + (out[2] * sc.grad_depth.to(DEV)).sum()
    leaves = (means, opa, col, rot, scales, theta, rho)
    loss.backward(retain_graph=True)
    first = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    loss.backward()
    for a, t in zip(first, leaves):
        assert a.abs().max() > 0
        assert ((t.grad - a).norm() / a.norm()).item() < 1e-5        # (float atomics: the order of the adds differs)


def test_capacity_mode_matches_exact_path(native_lib):
    """Sync-free forward (device-side instance count, capacity-sized scratch) gives the same images and
    gradients as the exact path; an under-sized capacity raises the overflow flag instead of writing out of bounds."""
    from monogs_amd import rasterizer as R
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(20000, "fr3_office", seed=131)
    st = _hip_st(sc)

    def run():
        leaf = lambda t: t.to(DEV).clone().requires_grad_(True)  # noqa: E731
        m, o, c, r = leaf(sc.means3D), leaf(sc.opacities), leaf(sc.colors), leaf(sc.rotations)
        s = leaf(sc.scales.repeat(1, 3))
        th = torch.zeros(3, device=DEV, requires_grad=True)
        out = GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), opacities=o, colors_precomp=c, scales=s,
                                     rotations=r, theta=th, rho=torch.zeros(3, device=DEV, requires_grad=True))
        ((out[0] * sc.grad_color.to(DEV)).sum() + (out[2] * sc.grad_depth.to(DEV)).sum()).backward()
        return [x.detach().clone() for x in out], m.grad.clone(), th.grad.clone()

    R.set_sync_free(False)
    out_a, gm_a, gt_a = run()                       # exact path: records the capacity hint
    try:
        R.set_sync_free(True, headroom=1.2)
        out_b, gm_b, gt_b = run()
        assert not R.check_overflow()
        for x, y in zip(out_a, out_b):
            assert torch.equal(x, y)
        # (two runs of the same backward differ by the order of its float atomics: absolute noise ~1e-7 of the largest entry)
        close = lambda a, b: torch.allclose(a, b, rtol=1e-4, atol=1e-6 * float(a.abs().max()))  # noqa: E731
        assert close(gm_a, gm_b) and close(gt_a, gt_b)
        # starve the capacity: the flag must fire, nothing may crash
        key = (20000, 640, 480)
        R._capacity_hint[key] = 1000
        R.set_sync_free(True, headroom=1.0)
        run()
        assert R.check_overflow()
        assert R._capacity_hint[key] >= 2000
    finally:
        R.set_sync_free(False)
        R._capacity_hint.pop((20000, 640, 480), None)


def test_radix_lookback_timeout_is_reported_in_every_mode(native_lib):
    """The radix sort bounds its inter-workgroup waits; a wait that runs out must surface instead of silently
    producing a wrong blend order.  With the spin bound forced to 0 (mgs_debug_set_radix_spin_limit) every tile that
    has to wait at all gives up: the exact path reports the depth sort at its count read-back, the capacity / graph
    path folds both sorts' flags into the status word that check_overflow() reads."""
    from monogs_amd import _lib
    from monogs_amd import rasterizer as R
    from monogs_amd.rasterizer import GaussianRasterizer
    lib = _lib.load()
    sc = make_scene(150000, "fr3_office", seed=77)       # ~150 tiles of 1024 keys in the depth sort: look-back is certain
    st = _hip_st(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731
    args = dict(means3D=dev(sc.means3D), means2D=torch.zeros(150000, 3, device=DEV), opacities=dev(sc.opacities),
                colors_precomp=dev(sc.colors), scales=dev(sc.scales), rotations=dev(sc.rotations))
    R.check_overflow()
    R.set_sync_free(False)
    ref = GaussianRasterizer(st)(**args)                 # healthy run: records the capacity hint
    assert not R.check_overflow()
    try:
        assert lib.mgs_debug_set_radix_spin_limit(0) == 0
        with pytest.raises(_lib.MonoGSNativeError, match="look-back"):
            GaussianRasterizer(st)(**args)               # exact path: depth-sort flag read at the existing sync
        R._pending_overflow.clear()
        R.set_sync_free(True)
        GaussianRasterizer(st)(**args)                   # capacity path: no sync inside ...
        with pytest.raises(RuntimeError, match="timed out"):
            R.check_overflow()                           # ... the status word carries the sorts' flags
    finally:
        lib.mgs_debug_set_radix_spin_limit(0xFFFFFFFF)
        R.set_sync_free(False)
        R._pending_overflow.clear()
    out = GaussianRasterizer(st)(**args)                 # bound restored: clean again, same image
    assert not R.check_overflow()
    assert torch.equal(out[0], ref[0])


def test_scan_paths_agree(native_lib):
    """The single-launch scan of SLAM-sized maps (block totals all-gathered through status words) and the two-launch scan of
    large ones give the same offsets: tables and image bit for bit, in the exact and in the capacity mode."""
    from monogs_amd import _lib
    from monogs_amd import rasterizer as R
    from monogs_amd.debug import forward_tables
    from monogs_amd.rasterizer import GaussianRasterizer
    lib = _lib.load()
    sc = make_scene(70000, "fr3_office", seed=9)          # 35 scan blocks
    st = _hip_st(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731
    args = dict(colors_precomp=dev(sc.colors), scales=dev(sc.scales), rotations=dev(sc.rotations))
    one = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args)
    full = dict(means3D=dev(sc.means3D), means2D=torch.zeros(70000, 3, device=DEV), opacities=dev(sc.opacities), **args)
    R.set_sync_free(True)
    try:
        with torch.no_grad():
            cap_one = GaussianRasterizer(st)(**full)
        lib.mgs_debug_set_option(b"scan_small", 0)
        two = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args)
        with torch.no_grad():
            cap_two = GaussianRasterizer(st)(**full)
        assert not R.check_overflow()
    finally:
        lib.mgs_debug_set_option(b"scan_small", -1)
        R.set_sync_free(False)
    assert one["num_rendered"] == two["num_rendered"] > 0
    for k in ("ranges", "point_list", "tile_sorted", "color", "n_contrib", "n_touched"):
        assert torch.equal(one[k], two[k]), k
    assert torch.equal(cap_one[0], cap_two[0]) and torch.equal(cap_one[0], one["color"])


@pytest.mark.parametrize("n,intr", [(1, "fr3_office"), (63, "fr3_office"), (1025, "fr3_office"), (9000, "replica"), (24576, "fr3_office")])
def test_small_depth_chain_matches_the_multi_launch_path(native_lib, n, intr):
    """The whole depth chain of a small map -- four stable radix passes, the rectangle gather, the prefix sum of tiles touched --
    as ONE single-workgroup launch with everything in LDS (depth_chain_small_kernel, mgs_debug_set_option("depth_small", 1):
    measured slower than the six launches it replaces and therefore not the default).  Same keys, same stable
    order: the depth order, the per-Gaussian tables, the per-tile lists and the image equal the multi-launch path's bit for bit,
    in the exact and in the capacity mode; depth ties (duplicated Gaussians) keep their index order in both."""
    from monogs_amd import _lib
    from monogs_amd import rasterizer as R
    from monogs_amd.debug import forward_tables
    from monogs_amd.rasterizer import GaussianRasterizer
    lib = _lib.load()
    sc = make_scene(n, intr, seed=12)
    st = _hip_st(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731
    means, opac, cols, scales, rots = dev(sc.means3D), dev(sc.opacities), dev(sc.colors), dev(sc.scales), dev(sc.rotations)
    if n >= 1025:                                  # exact depth ties: every 7th Gaussian is a copy of its left neighbour's position
        means = means.clone()
        means[7::7] = means[6:-1:7][:means[7::7].shape[0]]
    args = dict(colors_precomp=cols, scales=scales, rotations=rots)
    multi = forward_tables(st, means, opac, **args)
    full = dict(means3D=means, means2D=torch.zeros(n, 3, device=DEV), opacities=opac, **args)
    R.set_sync_free(True)
    try:
        with torch.no_grad():
            cap_multi = GaussianRasterizer(st)(**full)
        lib.mgs_debug_set_option(b"depth_small", 1)
        small = forward_tables(st, means, opac, **args)
        with torch.no_grad():
            cap_small = GaussianRasterizer(st)(**full)
        assert not R.check_overflow()
    finally:
        lib.mgs_debug_set_option(b"depth_small", 0)
        R.set_sync_free(False)
    assert small["num_rendered"] == multi["num_rendered"]
    for k in ("perm", "tiles_touched", "ranges", "point_list", "tile_sorted", "color", "n_contrib", "n_touched", "radii"):
        assert torch.equal(small[k], multi[k]), k
    assert torch.equal(cap_small[0], cap_multi[0]) and torch.equal(cap_small[0], small["color"])
    # and the order IS the contract's: ascending (depth bits, index) over the visible Gaussians, the culled ones behind them
    vis = small["radii"] > 0
    key = torch.where(vis, small["depth_key"].long() & 0xFFFFFFFF, torch.full_like(small["depth_key"].long(), 0xFFFFFFFF))
    ref = torch.sort(key, stable=True).indices
    assert torch.equal(small["perm"].long(), ref)


@pytest.mark.parametrize("n,intr", [(3000, "fr3_office"), (70000, "fr3_office"), (40000, "replica")])
def test_duplicate_emission_paths_agree(native_lib, n, intr):
    """duplicate_kernel emits the (tile, Gaussian) instances either Gaussian-major (a wave owns the slots of its 64
    Gaussians) or slot-major (every wave owns an equal share of the output slots and searches its first Gaussian): same
    keys, same values, hence the same tables and image, in the exact and in the capacity mode."""
    from monogs_amd import _lib
    from monogs_amd import rasterizer as R
    from monogs_amd.debug import forward_tables
    from monogs_amd.rasterizer import GaussianRasterizer
    lib = _lib.load()
    sc = make_scene(n, intr, seed=13)
    st = _hip_st(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731
    args = dict(colors_precomp=dev(sc.colors), scales=dev(sc.scales), rotations=dev(sc.rotations))
    full = dict(means3D=dev(sc.means3D), means2D=torch.zeros(n, 3, device=DEV), opacities=dev(sc.opacities), **args)
    out = {}
    try:
        for mode in (0, 1):
            lib.mgs_debug_set_option(b"dup_slot_major", mode)
            out[mode] = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args)
            R.set_sync_free(True)
            with torch.no_grad():
                GaussianRasterizer(st)(**full)                                  # (records the capacity hint)
                out[mode]["cap"] = GaussianRasterizer(st)(**full)[0]
            assert not R.check_overflow()
            R.set_sync_free(False)
    finally:
        lib.mgs_debug_set_option(b"dup_slot_major", -1)
        R.set_sync_free(False)
    assert out[0]["num_rendered"] == out[1]["num_rendered"] > 0
    for k in ("ranges", "point_list", "tile_sorted", "color", "n_contrib", "n_touched"):
        assert torch.equal(out[0][k], out[1][k]), k
    assert torch.equal(out[0]["cap"], out[1]["cap"]) and torch.equal(out[0]["cap"], out[0]["color"])


def test_exact_path_reports_an_earlier_forwards_status_at_its_own_read_back(native_lib):
    """The default (exact) path never synchronises for the tile sort's status word: the NEXT exact forward collects it at
    the count read-back it performs anyway (`mgs_forward_preprocess(prev_status, prev_status_out)`), so a look-back timeout
    there surfaces one forward later instead of never (debug=False, nobody calling check_overflow)."""
    from monogs_amd import rasterizer as R
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(3000, "fr3_office", seed=9)
    st = _hip_st(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731
    args = dict(means3D=dev(sc.means3D), means2D=torch.zeros(3000, 3, device=DEV), opacities=dev(sc.opacities),
                colors_precomp=dev(sc.colors), scales=dev(sc.scales), rotations=dev(sc.rotations))
    R.check_overflow()
    R.set_sync_free(False)
    ref = GaussianRasterizer(st)(**args)
    assert R._State.exact_pending is not None and int(R._State.exact_pending[1].item()) == 0
    R._State.exact_pending[1].fill_(R.STATUS_TILE_SORT_TIMEOUT)      # what ranges_kernel leaves after a timed-out tile sort
    with pytest.raises(RuntimeError, match="tile sort"):
        GaussianRasterizer(st)(**args)                                # raised by the next forward, at ITS synchronisation
    assert R._State.exact_pending is None
    out = GaussianRasterizer(st)(**args)                              # and only once
    assert torch.equal(out[0], ref[0])
    assert not R.check_overflow()
    # the same word through check_overflow()
    GaussianRasterizer(st)(**args)
    R._State.exact_pending[1].fill_(R.STATUS_TILE_SORT_TIMEOUT)
    with pytest.raises(RuntimeError, match="tile sort"):
        R.check_overflow()
    assert not R.check_overflow()


def test_c_abi_renders_the_background_for_an_empty_map_in_capacity_mode(native_lib):
    """P == 0 through the C ABI with geometry == NULL and a non-zero capacity (the binding the reference's FFI would use has
    no Python guard in front of it): nothing was preprocessed, so nothing may be sorted -- background image, no fault."""
    import ctypes as C
    from monogs_amd import _lib
    from monogs_amd.rasterizer import _camera, _stream
    lib = _lib.load()
    sc = make_scene(10, "fr3_office", seed=1, bg=(0.1, 0.2, 0.3))
    st = _hip_st(sc)
    keep = []
    cam = _camera(st, 0, keep, 3)
    H, W = st.image_height, st.image_width
    u8 = dict(dtype=torch.uint8, device=DEV)
    img = torch.empty(lib.mgs_image_bytes(W, H), **u8)
    binning = torch.empty(lib.mgs_binning_bytes(100000, W, H), **u8)
    out = torch.full((5, H, W), -1.0, device=DEV)
    status = torch.full((1,), 77, dtype=torch.int32, device=DEV)
    for fn, r in ((lib.mgs_forward_render_capacity, 100000), (lib.mgs_forward_render, 0)):
        out.fill_(-1.0)
        _lib.check(fn(C.byref(cam), 0, r, None, binning.data_ptr(), img.data_ptr(), out[0:3].data_ptr(), out[3:4].data_ptr(),
                      out[4:5].data_ptr(), None, status.data_ptr(), None, _stream()), "forward render, P = 0")
        torch.cuda.synchronize()
        assert torch.allclose(out[0:3], torch.tensor([0.1, 0.2, 0.3], device=DEV)[:, None, None].expand(3, H, W))
        assert float(out[3:].abs().max()) == 0.0


def test_tile_sort_timeout_raises_its_own_status_bit_and_blends_nothing(native_lib):
    """A look-back timeout in the TILE sort (the depth sort healthy): the status word carries MGS_STATUS_TILE_SORT_TIMEOUT,
    the tile ranges stay empty and nothing is blended -- no instance index is ever read from the half-written list."""
    from monogs_amd import _lib
    from monogs_amd.debug import forward_tables
    lib = _lib.load()
    sc = make_scene(60000, "fr3_office", seed=5)
    st = _hip_st(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731
    args = dict(colors_precomp=dev(sc.colors), scales=dev(sc.scales), rotations=dev(sc.rotations))
    lib.mgs_debug_set_option(b"radix_scanned", 0)         # one-sweep tile sort whatever the instance count
    try:
        good = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args)
        assert good["status"] == 0 and float(good["opacity"].max()) > 0.5 and good["num_rendered"] > 8192
        bad = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args,
                             between=lambda: lib.mgs_debug_set_radix_spin_limit(0))
    finally:
        lib.mgs_debug_set_radix_spin_limit(0xFFFFFFFF)
        lib.mgs_debug_set_option(b"radix_scanned", -1)
    assert bad["status"] == 4                             # MGS_STATUS_TILE_SORT_TIMEOUT only: the depth sort had finished
    assert float(bad["opacity"].abs().max()) == 0.0 and int(bad["n_contrib"].max()) == 0
    assert int(bad["ranges"].abs().max()) == 0


@pytest.mark.parametrize("n,intr", [(3000, "fr3_office"), (60000, "fr3_office"), (40000, "replica")])
def test_ticket_and_block_id_tile_ids_agree(native_lib, n, intr):
    """Every one-sweep radix pass hands out its tile ids by an atomic ticket (placement-independent forward progress beside
    other grids); a caller that declares the device its own (`MGS_FLAG_EXCLUSIVE_DEVICE`, `rasterizer.exclusive_device()`)
    gets block ids for sorts of at most 256 tiles.  Same tables, same image, bit for bit, exact and capacity mode."""
    from monogs_amd import rasterizer as R
    from monogs_amd.debug import forward_tables
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(n, intr, seed=21)
    st = _hip_st(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731
    args = dict(colors_precomp=dev(sc.colors), scales=dev(sc.scales), rotations=dev(sc.rotations))
    full = dict(means3D=dev(sc.means3D), means2D=torch.zeros(n, 3, device=DEV), opacities=dev(sc.opacities), **args)
    out = {}
    try:
        for excl in (False, True):
            with R.exclusive_device(excl):
                out[excl] = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args)
                R.set_sync_free(True)
                with torch.no_grad():
                    out[excl]["cap"] = GaussianRasterizer(st)(**full)[0]
                assert not R.check_overflow()
                R.set_sync_free(False)
    finally:
        R.set_sync_free(False)
    assert out[False]["num_rendered"] == out[True]["num_rendered"] > 0 and out[False]["status"] == out[True]["status"] == 0
    for k in ("ranges", "point_list", "perm", "color", "n_contrib", "n_touched"):
        assert torch.equal(out[False][k], out[True][k]), k
    assert torch.equal(out[False]["cap"], out[True]["cap"]) and torch.equal(out[False]["cap"], out[False]["color"])


def test_blend_event_hook_times_the_two_blend_kernels(native_lib):
    """mgs_debug_set_blend_events: the library records the caller's events right around the blend-forward and blend-backward
    launches (what bench.py times the dominant kernel with inside its timed region) -- no sync, results unchanged, and
    nothing is recorded once the hook is off again."""
    from monogs_amd._lib import check
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(20000, "fr3_office", seed=4)
    st = _hip_st(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731

    def run():
        xyz = dev(sc.means3D).requires_grad_(True)
        out = GaussianRasterizer(st)(means3D=xyz, means2D=torch.zeros(20000, 3, device=DEV), opacities=dev(sc.opacities),
                                     colors_precomp=dev(sc.colors), scales=dev(sc.scales), rotations=dev(sc.rotations))
        torch.autograd.backward([out[0], out[2]], [dev(sc.grad_color), dev(sc.grad_depth)])
        return out[0].detach().clone(), xyz.grad.clone()

    ref = run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    assert native_lib.mgs_debug_set_blend_events(ev[0].cuda_event, None, None, None) == 1        # half a pair
    check(native_lib.mgs_debug_set_blend_events(*[e.cuda_event for e in ev]), "mgs_debug_set_blend_events")
    try:
        got = run()
    finally:
        check(native_lib.mgs_debug_set_blend_events(None, None, None, None), "mgs_debug_set_blend_events")
    torch.cuda.synchronize()
    t_f, t_b = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])
    assert 0.001 < t_f < 5.0 and 0.001 < t_b < 5.0, (t_f, t_b)
    assert torch.equal(got[0], ref[0])
    run()                                            # hook off: the events keep their times
    torch.cuda.synchronize()
    assert ev[0].elapsed_time(ev[1]) == t_f and ev[2].elapsed_time(ev[3]) == t_b


def test_exact_status_words_of_other_streams_wait_for_check_overflow(native_lib):
    """The exact path hands the previous forward's status word to the next forward's count read-back only when both ran on the
    same stream of the same device (the read-back is ordered behind that stream's kernels only); a word written on another
    stream is parked and read by check_overflow(), which drains the devices involved first."""
    from monogs_amd import rasterizer as R
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(4000, "fr3_office", seed=5)
    st = _hip_st(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731
    args = dict(means3D=dev(sc.means3D), means2D=torch.zeros(4000, 3, device=DEV), opacities=dev(sc.opacities),
                colors_precomp=dev(sc.colors), scales=dev(sc.scales), rotations=dev(sc.rotations))
    R.check_overflow()
    R.set_sync_free(False)
    side = torch.cuda.Stream()
    with torch.no_grad():
        a = GaussianRasterizer(st)(**args)                       # current stream: its word is pending
        assert R._State.exact_pending is not None and not R._State.exact_other
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            b = GaussianRasterizer(st)(**args)                   # another stream: the first word is parked, not read
        assert len(R._State.exact_other) == 1 and R._State.exact_pending is not None
        side.synchronize()
        c = GaussianRasterizer(st)(**args)                       # back on the first stream: the side stream's word is parked too
        assert len(R._State.exact_other) == 2
    assert not R.check_overflow()                                # reads all three after draining the device
    assert R._State.exact_pending is None and not R._State.exact_other
    assert torch.equal(a[0], b[0]) and torch.equal(a[0], c[0])


def test_streams_nograd_and_noncontiguous_inputs(native_lib):
    """The library launches on the caller's current stream, works under no_grad, with inputs that do not
    require grad and with non-contiguous views (made contiguous at the boundary, as upstream's .contiguous())."""
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(6000, "fr3_office", seed=141)
    st = _hip_st(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731
    args = dict(means3D=dev(sc.means3D), opacities=dev(sc.opacities), colors_precomp=dev(sc.colors),
                scales=dev(sc.scales.repeat(1, 3)), rotations=dev(sc.rotations))
    with torch.no_grad():
        ref = GaussianRasterizer(st)(means2D=torch.zeros_like(args["means3D"]), **args)
    # side stream
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        out = GaussianRasterizer(st)(means2D=torch.zeros_like(args["means3D"]), **args)
    s.synchronize()
    for a, b in zip(ref, out):
        assert torch.equal(a, b)
    # non-contiguous views of wider tensors
    wide = torch.zeros(6000, 8, device=DEV)
    wide[:, 1:4] = args["means3D"]
    colw = torch.zeros(6000, 6, device=DEV)
    colw[:, ::2] = args["colors_precomp"]
    a2 = dict(args, means3D=wide[:, 1:4], colors_precomp=colw[:, ::2])
    assert not a2["means3D"].is_contiguous() and not a2["colors_precomp"].is_contiguous()
    with torch.no_grad():
        out2 = GaussianRasterizer(st)(means2D=torch.zeros(6000, 3, device=DEV), **a2)
    for a, b in zip(ref, out2):
        assert torch.equal(a, b)
    # only the pose requires grad (tracking with a frozen map): Gaussian gradients are simply not produced
    th = torch.zeros(3, device=DEV, requires_grad=True)
    rh = torch.zeros(3, device=DEV, requires_grad=True)
    o3 = GaussianRasterizer(st)(means2D=torch.zeros(6000, 3, device=DEV), theta=th, rho=rh, **args)
    (o3[0] * sc.grad_color.to(DEV)).sum().backward()
    assert th.grad is not None and rh.grad is not None and th.grad.abs().sum() > 0
    _, og = rasterize_autograd(dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=sc.colors,
                                    scales=sc.scales.repeat(1, 3), rotations=sc.rotations),
                               scene_settings(sc, OracleSettings), sc.grad_color, torch.zeros_like(sc.grad_depth),
                               dtype=torch.float32)
    assert ((th.grad.cpu() - og["theta"]).norm() / og["theta"].norm()).item() < 1e-3
    assert ((rh.grad.cpu() - og["rho"]).norm() / og["rho"].norm()).item() < 1e-3


def test_graph_capture_of_forward_backward(native_lib):
    """A forward + backward captured in a hipGraph (capacity mode) replays to the same gradients."""
    from monogs_amd import rasterizer as R
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(8000, "fr3_office", seed=151)
    st = _hip_st(sc)
    m = sc.means3D.to(DEV).clone().requires_grad_(True)
    o, c, r = sc.opacities.to(DEV), sc.colors.to(DEV), sc.rotations.to(DEV)
    s3 = sc.scales.repeat(1, 3).to(DEV)
    gc, gd = sc.grad_color.to(DEV), sc.grad_depth.to(DEV)

    def step():
        out = GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), opacities=o, colors_precomp=c, scales=s3,
                                     rotations=r)
        ((out[0] * gc).sum() + (out[2] * gd).sum()).backward()
        return out[0]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        img_ref = step().detach().clone()        # eager: records the capacity hint
    torch.cuda.current_stream().wait_stream(side)
    g_ref = m.grad.clone()
    m.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        img = step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert not R.check_overflow()
    assert torch.equal(img, img_ref)
    # (replay vs eager differ by the order of the blend backward's float atomics: absolute noise ~1e-7 of the largest entry)
    assert torch.allclose(m.grad, g_ref, rtol=1e-4, atol=1e-6 * float(g_ref.abs().max()))
    R.clear_graph_flags()


def test_graph_replay_survives_eager_work_between_replays(native_lib):
    """Regression: the captured forward+backward must hold kernel nodes only.  With hipMemsetAsync / hipMemcpyAsync
    nodes inside, a replay faulted once eager copies had run between two replays (tools/graph_probe.py)."""
    from monogs_amd.rasterizer import GaussianRasterizer
    from monogs_amd import rasterizer as _r
    sc = make_scene(6000, "fr3_office", seed=3, device=DEV)
    rs = _hip_st(sc)
    m3d = sc.means3D.clone().requires_grad_(True)
    gc_ = torch.rand(3, rs.image_height, rs.image_width, device=DEV)

    def step():
        m3d.grad = None
        out = GaussianRasterizer(rs)(means3D=m3d, means2D=torch.zeros_like(m3d), opacities=sc.opacities,
                                     colors_precomp=sc.colors, scales=sc.scales, rotations=sc.rotations)
        (out[0] * gc_).sum().backward()
        return out[0]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ref = step().clone()
        gref = m3d.grad.clone()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        img = step()
    for k in range(4):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(img, ref), k
        assert (m3d.grad - gref).norm() <= 1e-5 * gref.norm(), k     # float atomics: summation order varies
        # eager traffic of the kinds that used to break the next replay
        img.flatten()[:2].tolist()
        scratch = img.clone()
        scratch.copy_(img)
        (img.double().cumsum(0).to(torch.int16))
    assert not _r.check_overflow()
    _r.clear_graph_flags()


def test_graph_scratch_is_not_recycled_by_eager_allocations(native_lib):
    """The lifetime half of the round-1 replay fault: every buffer the captured iteration
    writes (geometry / binning / image scratch, gradient accumulator, status word, outputs) is allocated DURING capture,
    i.e. from the graph's private pool, and must stay out of the eager allocator's hands while the graph lives --
    otherwise the kernel clears that replaced the faulting memset / memcpy nodes would now scribble over live eager
    tensors silently.  Checked without provoking anything: after capture, a burst of eager allocations of many sizes
    never lands inside a private-pool segment, before or after replays."""
    from monogs_amd.rasterizer import GaussianRasterizer
    from monogs_amd import rasterizer as _r
    sc = make_scene(6000, "fr3_office", seed=4, device=DEV)
    rs = _hip_st(sc)
    m3d = sc.means3D.clone().requires_grad_(True)
    gc_ = torch.rand(3, rs.image_height, rs.image_width, device=DEV)

    def step():
        m3d.grad = None
        out = GaussianRasterizer(rs)(means3D=m3d, means2D=torch.zeros_like(m3d), opacities=sc.opacities,
                                     colors_precomp=sc.colors, scales=sc.scales, rotations=sc.rotations)
        (out[0] * gc_).sum().backward()
        return out[0]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        img = step()

    def private_segments():
        return [(sg["address"], sg["address"] + sg["total_size"]) for sg in torch.cuda.memory_snapshot()
                if tuple(sg.get("segment_pool_id", (0, 0))) != (0, 0)]

    segs = private_segments()
    assert segs, "the capture allocated nothing from a private pool?"
    inside = lambda t: any(a <= t.data_ptr() < b for a, b in segs)  # noqa: E731
    assert inside(img) and inside(m3d.grad)            # outputs of the captured iteration live in the graph's pool
    for round_ in range(3):
        g.replay()
        torch.cuda.synchronize()
        eager = [torch.empty(n, dtype=torch.uint8, device=DEV) for n in (4, 24, 512, 4096, 1 << 16, 1 << 20, 6000 * 64, 1 << 24)]
        eager += [img.clone(), img.flatten()[:2].cpu().to(DEV)]
        assert not any(inside(t) for t in eager), round_
        assert private_segments() == segs              # and the pool itself neither moved nor shrank
        del eager
    assert not _r.check_overflow()
    _r.clear_graph_flags()


def test_pose_only_backward_matches_full(native_lib):
    """When colours / opacities take no gradient the blend backward reduces 6 sums instead of 10; the geometry and pose
    gradients must be the ones of the full path."""
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(20000, "fr3_office", seed=7, device=DEV)
    st = _hip_st(sc)
    gcol, gdep = sc.grad_color.to(DEV), sc.grad_depth.to(DEV)

    def run(full):
        m = sc.means3D.clone().requires_grad_(True)
        s_ = sc.scales.repeat(1, 3).clone().requires_grad_(True)
        r_ = sc.rotations.clone().requires_grad_(True)
        col = sc.colors.clone().requires_grad_(full)
        opa = sc.opacities.clone().requires_grad_(full)
        th = torch.zeros(3, device=DEV, requires_grad=True)
        rh = torch.zeros(3, device=DEV, requires_grad=True)
        out = GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), opacities=opa, colors_precomp=col,
                                     scales=s_, rotations=r_, theta=th, rho=rh)
        ((out[0] * gcol).sum() + (out[2] * gdep).sum()).backward()
        return [m.grad, s_.grad, r_.grad, th.grad, rh.grad]

    for a, b in zip(run(True), run(False)):
        assert (a - b).norm() <= 2e-5 * b.norm() + 1e-12


def test_isotropic_scales_match_repeat(native_lib):
    """scales [P,1] (mgs_camera.scale_dim = 1) == what render() gets from scales.repeat(1, 3), forward and backward."""
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(8000, "fr3_office", seed=9, device=DEV)
    st = _hip_st(sc)
    gcol, gdep = sc.grad_color.to(DEV), sc.grad_depth.to(DEV)

    def run(iso):
        m = sc.means3D.clone().requires_grad_(True)
        s1 = sc.scales.clone().requires_grad_(True)                     # [P,1]
        r_ = sc.rotations.clone().requires_grad_(True)
        out = GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), opacities=sc.opacities, colors_precomp=sc.colors,
                                     scales=s1 if iso else s1.repeat(1, 3), rotations=r_)
        ((out[0] * gcol).sum() + (out[2] * gdep).sum()).backward()
        return out, [m.grad, s1.grad, r_.grad]

    (oa, ga), (ob, gb) = run(True), run(False)
    for x, y in zip(oa, ob):
        assert torch.equal(x, y)
    assert ga[1].shape == (8000, 1)
    for a, b in zip(ga, gb):
        assert (a - b).norm() <= 2e-5 * b.norm() + 1e-12


def test_backward_walk_statistics_are_consistent(native_lib):
    """mgs_debug_blend_stats (the counting twin of the blend backward that feeds bench.py's `roofline.valu`): its counts obey
    the structure of the walk, and the number of active (pixel, instance) pairs equals what the ORACLE blends -- every
    pair the forward blended is replayed by the backward, no more, no less."""
    from monogs_amd.rasterizer import GaussianRasterizer, debug_blend_stats
    sc = make_scene(6000, "fr3_office", seed=12, mean_radius_px=8.0)
    m = sc.means3D.to(DEV).clone().requires_grad_(True)
    out = GaussianRasterizer(_hip_st(sc))(means3D=m, means2D=torch.zeros_like(m), opacities=sc.opacities.to(DEV),
                                          colors_precomp=sc.colors.to(DEV), scales=sc.scales.to(DEV),
                                          rotations=sc.rotations.to(DEV))
    st = debug_blend_stats(out[0])
    assert st["steps"] > 0 and 0 < st["active_survivors"] <= st["survivors"]
    assert st["active_survivors"] <= st["active_pairs"] <= 64 * st["active_survivors"]
    assert st["active_le2"] <= st["active_le4"] <= st["active_le8"] <= st["active_survivors"]
    assert st["inactive_by_depth_order"] <= st["survivors"] - st["active_survivors"]
    # oracle: pairs with a non-zero blend weight = sum over instances of the pixels they were blended into
    o = rasterize(sc.means3D, None, sc.opacities, scene_settings(sc, OracleSettings), colors_precomp=sc.colors,
                  scales=sc.scales.repeat(1, 3), rotations=sc.rotations, want_ambiguous=True)
    geom, pl, rg = o.aux["geom"], o.aux["point_list"], o.aux["ranges"]
    H, W = 480, 640
    pairs = 0
    n_contrib = o.aux["n_contrib"][0]
    for t in range(rg.shape[0]):
        s, e = int(rg[t, 0]), int(rg[t, 1])
        if e <= s:
            continue
        tx, ty = t % 40, t // 40
        ys, xs = torch.meshgrid(torch.arange(ty * 16, min(ty * 16 + 16, H)), torch.arange(tx * 16, min(tx * 16 + 16, W)), indexing="ij")
        px, py = xs.reshape(-1).float(), ys.reshape(-1).float()
        ids = pl[s:e]
        xy, con, op = geom["xy"][ids], geom["conic"][ids], geom["opacity"][ids]
        dx, dy = xy[None, :, 0] - px[:, None], xy[None, :, 1] - py[:, None]
        power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
        alpha = torch.clamp_max(op[None, :] * torch.exp(power), 0.99)
        last = n_contrib[ys.reshape(-1), xs.reshape(-1)]
        k = torch.arange(1, e - s + 1)[None, :]
        pairs += int(((k <= last[:, None]) & ~(power > 0) & ~(alpha < 1.0 / 255.0)).sum())
    amb = int(o.aux["ambiguous"].sum())
    assert abs(st["active_pairs"] - pairs) <= 64 * amb + 8, (st["active_pairs"], pairs, amb)
