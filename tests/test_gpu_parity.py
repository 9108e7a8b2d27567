"""HIP path vs the CPU oracle on seeded scenes (run on the MI355X box: pytest -m gpu).

Bars (BASELINE.json north_star): tile ranges / radii / blend order bit-exact; pixel Linf <= 1e-4;
gradient rtol <= 1e-3.  Pixels whose blend contains a decision within a few float32 ulps of its
threshold (alpha vs 1/255, T vs 1e-4 / 0.5) are reported and exempted from the pixel bar: a
different but equally valid exp rounding flips them, upstream included.
"""
import pytest
import torch

from monogs_amd.synthetic import make_scene, scene_settings
from oracle import OracleSettings, rasterize, rasterize_autograd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hip_settings(sc):
    from monogs_amd.rasterizer import GaussianRasterizationSettings
    return scene_settings(sc, GaussianRasterizationSettings, device=DEV)


def _inputs(sc, aniso=False):
    scales = sc.scales if sc.scales.shape[1] == 3 else sc.scales.repeat(1, 3)
    return dict(means3D=sc.means3D, opacities=sc.opacities, colors_precomp=sc.colors, scales=scales,
                rotations=sc.rotations)


def _run_hip(sc, inp, st):
    from monogs_amd.rasterizer import GaussianRasterizer
    leaves = {k: v.to(DEV).clone().requires_grad_(True) for k, v in inp.items()}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    theta = torch.zeros(3, device=DEV, requires_grad=True)
    rho = torch.zeros(3, device=DEV, requires_grad=True)
    out = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                 shs=leaves.get("shs"), colors_precomp=leaves.get("colors_precomp"),
                                 scales=leaves.get("scales"), rotations=leaves.get("rotations"),
                                 cov3D_precomp=leaves.get("cov3D_precomp"), theta=theta, rho=rho)
    color, radii, depth, opacity, n_touched = out
    loss = (color * sc.grad_color.to(DEV)).sum() + (depth * sc.grad_depth.to(DEV)).sum()
    loss.backward()
    grads = {k: v.grad.cpu() for k, v in leaves.items()}
    grads.update(means2D=means2D.grad.cpu(), theta=theta.grad.cpu(), rho=rho.grad.cpu())
    return [o.detach().cpu() for o in out], grads


def _check_grads(grads, ograds, rtol=1e-3, max_outlier_frac=2e-4, l2_tol=1e-4):
    """north_star: gradient rtol <= 1e-3.  Held here as (i) relative L2 error of every gradient tensor <= 1e-4 (measured:
    ~1e-6), ten times inside the bar, and (ii) ELEMENTWISE |got - ref| <= 1e-3 |ref| + 1e-5 max|ref| for all but a
    2e-4 fraction of the elements: a pixel whose alpha or transmittance sits within an ulp of a threshold may take
    the other branch on the GPU (different exp / rounding), which moves the handful of gradient entries it feeds."""
    report = {}
    for k, ref in ograds.items():
        got = grads[k].reshape(ref.shape).double()
        ref = ref.double()
        scale = ref.abs().max().item()
        if scale == 0:
            assert got.abs().max().item() == 0, k
            continue
        rel_l2 = ((got - ref).norm() / ref.norm()).item()
        bad = ((got - ref).abs() > rtol * ref.abs() + 1e-5 * scale)
        report[k] = (rel_l2, bad.float().mean().item())
        assert rel_l2 <= l2_tol, f"{k}: relative L2 error {rel_l2:.3e}"
        assert bad.float().mean().item() <= max_outlier_frac, f"{k}: {bad.sum().item()} elements off"
    return report


@pytest.mark.parametrize("P,intr,seed,aniso", [(5000, "fr3_office", 0, False), (3000, "fr3_office", 5, True),
                                               (20000, "replica", 7, False)])
def test_forward_tables_bit_exact(native_lib, P, intr, seed, aniso):
    from monogs_amd.debug import forward_tables
    sc = make_scene(P, intr, seed=seed, anisotropic=aniso)
    inp = _inputs(sc)
    o = rasterize(sc.means3D, None, sc.opacities, scene_settings(sc, OracleSettings),
                  colors_precomp=sc.colors, scales=inp["scales"], rotations=sc.rotations, want_ambiguous=True)
    t = forward_tables(_hip_settings(sc), sc.means3D.to(DEV), sc.opacities.to(DEV),
                       colors_precomp=sc.colors.to(DEV), scales=inp["scales"].to(DEV),
                       rotations=sc.rotations.to(DEV))
    assert t["num_rendered"] == o.aux["num_rendered"]
    assert torch.equal(t["radii"].cpu(), o.radii)
    assert torch.equal(t["tiles_touched"].cpu().long(), o.aux["geom"]["tiles_touched"])
    assert torch.equal(t["ranges"].cpu().long(), o.aux["ranges"])
    assert torch.equal(t["point_list"].cpu().long(), o.aux["point_list"])
    keys = torch.from_numpy(o.aux["keys"].astype("int64"))          # tile << 32 | depth bits, sorted
    assert torch.equal(t["tile_sorted"].cpu().long(), keys >> 32)
    # the depth bits of every instance, read back through the blend order, reproduce the oracle's keys
    dk = t["depth_key"].cpu().long() & 0xFFFFFFFF
    assert torch.equal(dk[t["point_list"].cpu().long()], keys & 0xFFFFFFFF)
    amb = o.aux["ambiguous"]
    ok = ~amb
    err = (t["color"].cpu() - o.color).abs().amax(0)
    print(f"ambiguous pixels: {int(amb.sum())} of {amb.numel()}; Linf clear {err[ok].max():.2e}, "
          f"Linf all {err.max():.2e}")
    assert amb.float().mean() < 1e-3
    assert err[ok].max() <= 1e-4
    assert (t["depth"].cpu() - o.depth).abs()[0][ok].max() <= 1e-4 * max(1.0, o.depth.max().item())
    assert (t["opacity"].cpu() - o.opacity).abs()[0][ok].max() <= 1e-4
    assert (t["final_T"].cpu() - o.aux["final_T"][0]).abs()[ok].max() <= 1e-4
    nc_bad = (t["n_contrib"].cpu().long() != o.aux["n_contrib"][0])
    assert not (nc_bad & ok).any()
    nt_diff = (t["n_touched"].cpu() - o.n_touched).abs()
    assert nt_diff.sum() <= 4 * amb.sum() + 2, f"n_touched differs by {nt_diff.sum()}"
    # invariants
    assert ((t["n_touched"] > 0) <= (t["radii"] > 0)).all()
    assert torch.allclose(t["opacity"][0], 1 - t["final_T"], atol=1e-6)


@pytest.mark.parametrize("P,intr,seed,aniso,bg", [(5000, "fr3_office", 0, False, (0., 0., 0.)),
                                                  (4000, "fr3_office", 11, True, (0.3, 0.5, 0.7))])
def test_backward_vs_oracle(native_lib, P, intr, seed, aniso, bg):
    sc = make_scene(P, intr, seed=seed, anisotropic=aniso, bg=bg)
    inp = _inputs(sc)
    outs, grads = _run_hip(sc, inp, _hip_settings(sc))
    oout, ograds = rasterize_autograd(inp, scene_settings(sc, OracleSettings), sc.grad_color, sc.grad_depth,
                                      dtype=torch.float32, want_ambiguous=True)
    ok = ~oout.aux["ambiguous"]
    assert (outs[0] - oout.color).abs().amax(0)[ok].max() <= 1e-4
    rep = _check_grads(grads, ograds)
    print({k: (f"{a:.2e}", f"{b:.1e}") for k, (a, b) in rep.items()})


def test_c2_100k(native_lib):
    """BASELINE config 2: 100k Gaussians, 640x480, fwd+bwd vs the oracle."""
    sc = make_scene(100000, "fr3_office", seed=1)
    inp = _inputs(sc)
    outs, grads = _run_hip(sc, inp, _hip_settings(sc))
    oout, ograds = rasterize_autograd(inp, scene_settings(sc, OracleSettings), sc.grad_color, sc.grad_depth,
                                      dtype=torch.float32, want_ambiguous=True)
    amb = oout.aux["ambiguous"]
    ok = ~amb
    err = (outs[0] - oout.color).abs().amax(0)
    print(f"C2: ambiguous {int(amb.sum())}; Linf clear {err[ok].max():.2e}; all {err.max():.2e}")
    assert torch.equal(outs[1], oout.radii)
    assert err[ok].max() <= 1e-4
    rep = _check_grads(grads, ograds)
    print({k: (f"{a:.2e}", f"{b:.1e}") for k, (a, b) in rep.items()})


def _c2_frame(color, depth, seed=1):
    """A seeded synthetic RGB-D "ground-truth" frame for BASELINE config 2 (upstream gradients come from
    ``get_loss_mapping``, /root/reference/utils/slam_utils.py:101-146): the rendered frame disturbed by smooth colour noise and
    5 % depth noise, with what a sensor frame has and a uniform random gradient lacks -- a masked-out image border and masked
    patches (``viewpoint.mask``: no colour gradient there), depth holes (``depth == 0``: no depth gradient there)."""
    import types
    g = torch.Generator().manual_seed(seed)
    H, W = depth.shape[-2:]
    low = torch.rand(1, 3, H // 16 + 1, W // 16 + 1, generator=g)
    noise = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)[0] - 0.5
    rgb = (color + 0.3 * noise).clamp(0, 1)
    d = depth[0] * (1 + 0.05 * torch.randn(H, W, generator=g))
    holes = torch.nn.functional.interpolate(torch.rand(1, 1, H // 8 + 1, W // 8 + 1, generator=g), size=(H, W))[0, 0] < 0.12
    d = torch.where(holes | (depth[0] <= 0), torch.zeros_like(d), d)
    mask = torch.ones(H, W, dtype=torch.bool)
    mask[:12], mask[-12:], mask[:, :12], mask[:, -12:] = False, False, False, False
    mask &= torch.nn.functional.interpolate(torch.rand(1, 1, H // 32 + 1, W // 32 + 1, generator=g), size=(H, W))[0, 0] > 0.1
    return types.SimpleNamespace(rgb=rgb, depth=d, mask=mask, exposure_a=torch.tensor([0.03]), exposure_b=torch.tensor([-0.02]))


def test_c2_100k_mapping_loss_gradients(native_lib):
    """BASELINE config 2 as SURVEY.md section 8d defines it: 100 k Gaussians, 640x480, "RGB-D single frame" = the upstream
    gradients are those of ``get_loss_mapping`` against a seeded synthetic RGB-D frame -- piecewise-constant +-lambda / N on
    the valid pixels, exactly ZERO on masked-out pixels and in depth holes -- not uniform noise.  The HIP side runs the whole
    chain (rasteriser forward -> fused loss -> loss.backward() -> rasteriser backward); the oracle takes the SAME upstream
    tensors (from the PyTorch loss mirror, which tests/test_golden.py pins to the reference's own outputs, evaluated on the
    HIP images) through its own autograd."""
    import types
    from monogs_amd import fused_losses
    from monogs_amd.rasterizer import GaussianRasterizer
    from oracle.slam_losses import get_loss_mapping as loss_ref
    sc = make_scene(100000, "fr3_office", seed=1)
    inp = _inputs(sc)
    st = _hip_settings(sc)
    leaves = {k: v.to(DEV).clone().requires_grad_(True) for k, v in inp.items()}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    theta = torch.zeros(3, device=DEV, requires_grad=True)
    rho = torch.zeros(3, device=DEV, requires_grad=True)
    color, radii, depth, opacity, n_touched = GaussianRasterizer(st)(
        means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], colors_precomp=leaves["colors_precomp"],
        scales=leaves["scales"], rotations=leaves["rotations"], theta=theta, rho=rho)
    vp_cpu = _c2_frame(color.detach().cpu(), depth.detach().cpu())
    vp = types.SimpleNamespace(rgb=vp_cpu.rgb.to(DEV), depth=vp_cpu.depth.to(DEV), mask=vp_cpu.mask.to(DEV),
                               exposure_a=vp_cpu.exposure_a.to(DEV).requires_grad_(True),
                               exposure_b=vp_cpu.exposure_b.to(DEV).requires_grad_(True))
    loss = fused_losses.get_loss_mapping(color, depth, vp)
    loss.backward()
    grads = {k: v.grad.cpu() for k, v in leaves.items()}
    grads.update(means2D=means2D.grad.cpu(), theta=theta.grad.cpu(), rho=rho.grad.cpu())
    # the same loss in plain PyTorch on the HIP images: value, upstream gradients, exposure gradients
    c_ref, d_ref = color.detach().cpu().requires_grad_(True), depth.detach().cpu().requires_grad_(True)
    vp_cpu.exposure_a.requires_grad_(True); vp_cpu.exposure_b.requires_grad_(True)
    l_ref = loss_ref(c_ref, d_ref, vp_cpu)
    g_color, g_depth, g_a, g_b = torch.autograd.grad(l_ref, [c_ref, d_ref, vp_cpu.exposure_a, vp_cpu.exposure_b])
    assert abs(float(loss.detach()) - float(l_ref)) <= 1e-5 * abs(float(l_ref))
    # (sums of ~10^6 signed terms of magnitude 1 / N that cancel to ~3e-4: compared on the scale of the terms, not of the sum)
    assert torch.allclose(vp.exposure_a.grad.cpu(), g_a, rtol=1e-3, atol=3e-6) and torch.allclose(vp.exposure_b.grad.cpu(), g_b, rtol=1e-3, atol=3e-6)
    # the upstream gradients the fused loss handed to the rasteriser (the same deterministic kernels, called again on the same
    # images) against the mirror's: piecewise constant +-k, so they either agree to rounding or differ by a whole k where the
    # residual rounds to zero in one evaluation and not in the other -- at most a handful of elements
    lg = fused_losses.loss_grads(color.detach(), depth.detach(), None, vp, tracking=False)
    f_color, f_depth = lg.d_render.cpu(), lg.d_depth.cpu()
    for got, ref in ((f_color, g_color), (f_depth, g_depth)):
        off = (got - ref).abs() > 1e-4 * ref.abs().max()
        assert int(off.sum()) <= 3, int(off.sum())
        assert torch.allclose(got[~off], ref[~off], rtol=1e-5, atol=0)
    # the upstream gradient has what C2 is about: zero regions and a piecewise-constant magnitude
    zero_c, zero_d = (f_color.abs().sum(0) == 0).float().mean().item(), (f_depth[0] == 0).float().mean().item()
    print(f"C2 (mapping loss): dL/dcolor zero on {zero_c:.1%} of the pixels, dL/ddepth zero on {zero_d:.1%}")
    assert 0.05 < zero_c < 0.6 and 0.05 < zero_d < 0.8
    # the oracle pulls back exactly the tensors the HIP backward received
    oout, ograds = rasterize_autograd(inp, scene_settings(sc, OracleSettings), f_color, f_depth, dtype=torch.float32,
                                      want_ambiguous=True)
    ok = ~oout.aux["ambiguous"]
    assert (color.detach().cpu() - oout.color).abs().amax(0)[ok].max() <= 1e-4
    # (a sign-valued upstream gradient makes every per-Gaussian sum a cancellation of equal-magnitude terms: float32 results sit
    #  ~1e-5 from a float64 evaluation where uniform noise gives ~1e-6 -- the oracle's as much as the kernels')
    rep = _check_grads(grads, ograds)
    print({k: (f"{a:.2e}", f"{b:.1e}") for k, (a, b) in rep.items()})
    # ... and against the float64 oracle the HIP gradients are no further off than the float32 oracle's own
    _, g64 = rasterize_autograd(inp, scene_settings(sc, OracleSettings), f_color, f_depth, dtype=torch.float64)
    for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp"):
        ref = g64[k].double()
        e_hip = ((grads[k].reshape(ref.shape).double() - ref).norm() / ref.norm()).item()
        e_o32 = ((ograds[k].reshape(ref.shape).double() - ref).norm() / ref.norm()).item()
        print(f"  {k}: vs float64 oracle -- HIP {e_hip:.2e}, float32 oracle {e_o32:.2e}")
        assert e_hip <= max(3.0 * e_o32, 1e-5), (k, e_hip, e_o32)
    # THE END-TO-END CHAIN, bounded instead of routed around: HIP forward -> FUSED loss -> HIP backward
    # against the oracle fed the PLAIN PYTORCH loss's upstream gradients (the mirror of /root/reference/utils/slam_utils.py:101-146
    # that tests/test_golden.py pins to the reference's own outputs).  The two upstream tensors differ in at most three of 921 600
    # elements -- an L1 residual that rounds to zero in one evaluation and not in the other flips that element's gradient by a whole
    # +-lambda / N -- and that is ALL that separates the chains: at north_star's own bar (relative L2 <= 1e-3 per tensor), measured
    # ~1.3e-4 for means3D where the like-for-like comparison above reads ~1e-5.
    _, ograds_mirror = rasterize_autograd(inp, scene_settings(sc, OracleSettings), g_color, g_depth, dtype=torch.float32)
    chain = {}
    for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp", "theta", "rho"):
        ref = ograds_mirror[k].double()
        chain[k] = ((grads[k].reshape(ref.shape).double() - ref).norm() / ref.norm()).item()
    print("C2 end-to-end chain (fused loss + HIP backward vs torch loss + oracle), relative L2:", {k: f"{v:.2e}" for k, v in chain.items()})
    assert all(v <= 1e-3 for v in chain.values()), chain


@pytest.mark.parametrize("pose_only", [False, True])
@pytest.mark.parametrize("P,intr,seed", [(5000, "fr3_office", 0), (100000, "fr3_office", 1), (60000, "replica", 4)])
def test_blend_backward_paths_agree(native_lib, P, intr, seed, pose_only):
    """The three blend backwards -- `blend_backward_s_kernel` (default since round 5: the record fetched one survivor ahead with
    two SGPR-offset scalar loads, the per-pixel side under a narrowed EXEC, one row reduction per four survivors),
    `blend_backward_t_kernel` (round 3: the same transposed accumulation, record fetched on demand, activity through v_cndmask;
    mgs_debug_set_option("blend_bwd_transposed", 1)) and `blend_backward_kernel` (option 0: one 64-lane reduction per survivor)
    -- form the same sums; the first two in the same order per batch, the third in another: every gradient agrees to 1e-6
    relative L2, in the ten-sum and in the six-sum (pose-only: the map takes no gradient) variant."""
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(P, intr, seed=seed)
    st = _hip_settings(sc)
    inp = {k: v.to(DEV) for k, v in _inputs(sc).items()}
    gc, gd = sc.grad_color.to(DEV), sc.grad_depth.to(DEV)

    def run():
        leaves = {k: (v.clone() if pose_only else v.clone().requires_grad_(True)) for k, v in inp.items()}
        theta = torch.zeros(3, device=DEV, requires_grad=True)
        rho = torch.zeros(3, device=DEV, requires_grad=True)
        m2 = torch.zeros_like(leaves["means3D"], requires_grad=not pose_only)
        out = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                     colors_precomp=leaves["colors_precomp"], scales=leaves["scales"],
                                     rotations=leaves["rotations"], theta=theta, rho=rho)
        torch.autograd.backward([out[0], out[2]], [gc, gd])
        g = dict(theta=theta.grad.clone(), rho=rho.grad.clone())
        if not pose_only:
            g.update({k: v.grad.clone() for k, v in leaves.items()}, means2D=m2.grad.clone())
        return g
    try:
        native_lib.mgs_debug_set_option(b"blend_bwd_transposed", 2)
        a = run()
        native_lib.mgs_debug_set_option(b"blend_bwd_transposed", 1)
        b1 = run()
        native_lib.mgs_debug_set_option(b"blend_bwd_transposed", 0)
        b0 = run()
    finally:
        native_lib.mgs_debug_set_option(b"blend_bwd_transposed", 2)
    for b in (b1, b0):
        assert set(a) == set(b) and len(a) == (2 if pose_only else 8)
        for k in a:
            x, y = a[k].double(), b[k].double()
            assert y.norm() > 0, k
            rel = ((x - y).norm() / y.norm()).item()
            assert rel <= (1e-5 if k in ("theta", "rho") else 1e-6), (k, rel)      # (6 floats summed over the whole map)


def test_knn(native_lib):
    from monogs_amd.knn import distCUDA2
    from oracle import dist2_knn
    g = torch.Generator().manual_seed(0)
    for P in (4, 77, 9600, 25500):
        pts = torch.rand(P, 3, generator=g) * 4 - 2
        got = distCUDA2(pts.to(DEV)).cpu()
        ref = dist2_knn(pts)
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-9), P
    # duplicates count with distance zero
    pts = torch.rand(100, 3, generator=g)
    pts[50:] = pts[:50]
    got = distCUDA2(pts.to(DEV)).cpu()
    assert torch.allclose(got, dist2_knn(pts), rtol=1e-5, atol=1e-9)


def test_c5_full_size_properties(native_lib, monkeypatch):
    """BASELINE config 5 (2 M Gaussians, 1920x1080) is too large for the oracle, so it is checked through
    size-independent properties: the tile lists partition [0, R) and are sorted by (tile, depth bits, index),
    the image invariants hold, the forward is bitwise repeatable, and the backward is linear in the upstream
    gradient."""
    from monogs_amd.debug import forward_tables
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = make_scene(2_000_000, "davis_1080p", seed=2)
    st = _hip_settings(sc)
    dev = lambda t: t.to(DEV)  # noqa: E731
    args = dict(colors_precomp=dev(sc.colors), scales=dev(sc.scales.repeat(1, 3)), rotations=dev(sc.rotations))
    t = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args)
    R = t["num_rendered"]
    assert R == int(t["tiles_touched"].long().sum())
    rg = t["ranges"].long()
    ne = rg[rg[:, 1] > rg[:, 0]]
    assert ne[0, 0] == 0 and ne[-1, 1] == R and bool((ne[1:, 0] == ne[:-1, 1]).all())
    # sortedness: (tile, depth bits, Gaussian index) strictly increasing along the instance list
    tile = t["tile_sorted"].long()
    pl = t["point_list"].long()
    depth = (t["depth_key"].long() & 0xFFFFFFFF)[pl]
    key = (tile << 32) | depth
    d = key[1:] - key[:-1]
    assert bool((d >= 0).all())
    same = d == 0
    assert bool((pl[1:][same] > pl[:-1][same]).all())
    # every instance sits in a tile its Gaussian's rectangle covers (the sorted tile ids are no longer materialised: `tile`
    # is derived from the ranges, so this checks ranges + blend order against the per-Gaussian geometry)
    gx, gy = (st.image_width + 15) // 16, (st.image_height + 15) // 16
    rec = t["rec"][pl]
    cl = lambda v, hi: v.clamp(0, hi)  # noqa: E731
    x0, x1 = cl(torch.trunc((rec[:, 0] - rec[:, 15]) / 16), gx), cl(torch.trunc((rec[:, 0] + rec[:, 15] + 15) / 16), gx)
    y0, y1 = cl(torch.trunc((rec[:, 1] - rec[:, 15]) / 16), gy), cl(torch.trunc((rec[:, 1] + rec[:, 15] + 15) / 16), gy)
    tx, ty = (tile % gx).float(), (tile // gx).float()
    assert bool(((tx >= x0) & (tx < x1) & (ty >= y0) & (ty < y1)).all())
    starts = torch.repeat_interleave(torch.arange(rg.shape[0], device=DEV), (rg[:, 1] - rg[:, 0]))
    assert torch.equal(starts, tile)
    # image invariants
    assert torch.allclose(t["opacity"][0], 1 - t["final_T"], atol=1e-6)
    assert bool(((t["n_touched"] > 0) <= (t["radii"] > 0)).all())
    assert bool((t["n_contrib"].long() <= (rg[:, 1] - rg[:, 0]).max()).all())
    assert bool(torch.isfinite(t["color"]).all()) and float(t["opacity"].max()) <= 1.0
    # forward is bitwise repeatable (no atomics on the pixel path)
    t2 = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args)
    assert torch.equal(t["color"], t2["color"]) and torch.equal(t["point_list"], t2["point_list"])
    assert torch.equal(t["n_touched"], t2["n_touched"])
    # the two radix-sort paths (pre-scanned offsets, the default at this size, and one-sweep look-back) agree bit for bit
    native_lib.mgs_debug_set_option(b"radix_scanned", 0)
    try:
        t3 = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args)
    finally:
        native_lib.mgs_debug_set_option(b"radix_scanned", -1)
    assert torch.equal(t["point_list"], t3["point_list"]) and torch.equal(t["ranges"], t3["ranges"])
    assert torch.equal(t["color"], t3["color"])
    # capacity mode (device-side instance count; what bench.py times) renders the same image as the exact path
    from monogs_amd import rasterizer as _r
    _r.set_sync_free(True)
    try:
        with torch.no_grad():
            cap = GaussianRasterizer(st)(means3D=dev(sc.means3D), means2D=torch.zeros_like(dev(sc.means3D)),
                                         opacities=dev(sc.opacities), **args)
        assert not _r.check_overflow()
    finally:
        _r.set_sync_free(False)
    assert torch.equal(cap[0], t["color"]) and torch.equal(cap[4], t["n_touched"])
    # backward: linear in the upstream gradient
    def grads(scale):
        m = dev(sc.means3D).clone().requires_grad_(True)
        th = torch.zeros(3, device=DEV, requires_grad=True)
        out = GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), opacities=dev(sc.opacities), theta=th,
                                     rho=torch.zeros(3, device=DEV, requires_grad=True), **args)
        ((out[0] * dev(sc.grad_color)).sum() * scale + (out[2] * dev(sc.grad_depth)).sum() * scale).backward()
        return m.grad, th.grad
    g1, t1 = grads(1.0)
    g3, t3 = grads(3.0)
    assert ((g3 - 3 * g1).norm() / (3 * g1).norm()).item() < 1e-5
    assert ((t3 - 3 * t1).norm() / (3 * t1).norm()).item() < 1e-4


def test_knn_morton_path(native_lib, monkeypatch):
    """The Morton-box path (large clouds) against the all-pairs sweep (bit-identical) and the cKDTree oracle."""
    import time
    from monogs_amd.knn import distCUDA2
    from oracle import dist2_knn
    g = torch.Generator().manual_seed(5)
    uniform = torch.rand(60000, 3, generator=g) * 6 - 3
    # depth-map-like sheet + tight clusters + exact duplicates + a far outlier
    sheet = torch.cat([torch.rand(30000, 2, generator=g) * 4, torch.rand(30000, 1, generator=g) * 0.01 + 2.0], 1)
    clusters = torch.randn(20000, 3, generator=g) * 1e-3 + torch.randint(0, 5, (20000, 1), generator=g).float()
    dup = uniform[:5000].clone()
    mixed = torch.cat([sheet, clusters, dup, uniform[:5000], torch.tensor([[1e4, -1e4, 3e3]])])
    for name, pts in (("uniform", uniform), ("mixed", mixed), ("tiny", uniform[:70]), ("same", torch.ones(300, 3))):
        d = pts.to(DEV)
        native_lib.mgs_debug_set_option(b"knn_grid_min", 4)            # force the Morton-box path
        grid = distCUDA2(d).cpu()
        native_lib.mgs_debug_set_option(b"knn_grid_min", 1 << 30)      # force the all-pairs sweep
        sweep = distCUDA2(d).cpu()
        assert torch.equal(grid, sweep), (name, (grid - sweep).abs().max())
        assert torch.allclose(grid, dist2_knn(pts), rtol=1e-5, atol=1e-9), name
    native_lib.mgs_debug_set_option(b"knn_grid_min", -1)
    # full size: 2 M points (the C5 map) in milliseconds; property check = every result is a true 3-NN mean on a sample
    big = (torch.rand(2_000_000, 3, generator=g) * 20 - 10).to(DEV)
    distCUDA2(big)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = distCUDA2(big)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"distCUDA2 2M points: {dt * 1e3:.2f} ms")
    sample = torch.randint(0, big.shape[0], (256,), generator=g).to(DEV)
    d2 = ((big[sample][:, None, :] - big[None, :, :]) ** 2).sum(-1)
    d2[torch.arange(256, device=DEV), sample] = float("inf")
    ref = d2.topk(3, dim=1, largest=False).values.mean(1)
    assert torch.allclose(res[sample], ref, rtol=1e-4, atol=1e-9)
