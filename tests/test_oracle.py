"""The oracle's own correctness: known-answer cases, invariants, float64 gradcheck and finite
differences of the pose Jacobian.  (The oracle is the checker for the HIP path, so it is pinned here
first; the reference ships no vectors for the rasteriser -- parity unpinned, see oracle/__init__.py.)"""
import math

import numpy as np
import pytest
import torch

from monogs_amd import camera as cam
from monogs_amd.synthetic import make_scene, scene_settings
from oracle import OracleSettings, gs_oracle, rasterize, rasterize_autograd

D = torch.float64


def _settings(W=64, H=48, fx=60.0, fy=60.0, cx=None, cy=None, R=None, t=None, bg=(0., 0., 0.), dtype=torch.float32):
    cx = W / 2 if cx is None else cx
    cy = H / 2 if cy is None else cy
    R = torch.eye(3) if R is None else R
    t = torch.zeros(3) if t is None else t
    m = cam.camera_matrices(R, t, fx, fy, cx, cy, W, H)
    return OracleSettings(H, W, m.tanfovx, m.tanfovy, torch.tensor(bg), 1.0, m.viewmatrix, m.projmatrix,
                          m.projmatrix_raw, 0, m.campos, False, False)


def _one(mean, scale, opacity, color, st, rot=(1., 0., 0., 0.), dtype=D):
    n = len(mean)
    return rasterize(torch.tensor(mean, dtype=dtype), None, torch.tensor(opacity, dtype=dtype).reshape(n, 1), st,
                     colors_precomp=torch.tensor(color, dtype=dtype), scales=torch.tensor(scale, dtype=dtype),
                     rotations=torch.tensor([rot] * n, dtype=dtype), dtype=dtype)


def test_single_gaussian_centre_pixel():
    """A Gaussian whose mean projects exactly onto pixel (32, 24): alpha there is its opacity."""
    W, H, f = 64, 48, 60.0
    st = _settings(W, H, f, f)
    # pixel centre p maps from ndc via ((ndc+1)*S-1)/2  =>  ndc = (2p+1)/S - 1 ; x_c = ndc * z * W/(2f)
    z = 2.0
    px, py = 32, 24
    xc = ((2 * px + 1) / W - 1) * z * (W / (2 * f))
    yc = ((2 * py + 1) / H - 1) * z * (H / (2 * f))
    out = _one([[xc, yc, z]], [[0.1, 0.1, 0.1]], [0.6], [[0.2, 0.5, 0.9]], st)
    g = out.aux["geom"]
    assert torch.allclose(g["xy"][0], torch.tensor([float(px), float(py)], dtype=D), atol=1e-9)
    # EWA: cov = s^2 J J^T + 0.3 I with J = [[f/z, 0, -f x/z^2], [0, f/z, -f y/z^2]]
    s2, a = 0.1 ** 2, f / z
    cxx = s2 * a * a * (1 + (xc / z) ** 2) + 0.3
    cyy = s2 * a * a * (1 + (yc / z) ** 2) + 0.3
    cxy = s2 * a * a * (xc / z) * (yc / z)
    # (the settings carry float32 matrices and a float32-derived FoV, hence 1e-5 and not 1e-12)
    assert abs(g["cov2D"][0, 0].item() - cxx) < 1e-5 and abs(g["cov2D"][0, 1].item() - cxy) < 1e-5
    assert abs(g["cov2D"][0, 2].item() - cyy) < 1e-5
    mid, det = 0.5 * (cxx + cyy), cxx * cyy - cxy * cxy
    lam = mid + math.sqrt(max(0.1, mid * mid - det))
    assert out.radii[0].item() == math.ceil(3 * math.sqrt(lam))
    assert abs(out.opacity[0, py, px].item() - 0.6) < 1e-12
    assert torch.allclose(out.color[:, py, px], 0.6 * torch.tensor([0.2, 0.5, 0.9], dtype=D))
    assert abs(out.depth[0, py, px].item() - 0.6 * z) < 1e-12
    # one pixel to the right: alpha = o * exp(-conic_a / 2),  conic = cov^-1
    a1 = 0.6 * math.exp(-0.5 * (cyy / det))
    assert abs(out.opacity[0, py, px + 1].item() - a1) < 1e-7
    # counted where T(1-alpha) > 0.5, i.e. (T = 1) where 0 < alpha < 0.5
    assert out.n_touched[0].item() == int(((out.opacity[0] > 0) & (out.opacity[0] < 0.5)).sum())
    assert (out.aux["n_contrib"][0][out.opacity[0] > 0] == 1).all()


def test_two_gaussians_blend_in_depth_order_and_alpha_clamp():
    W, H, f = 64, 48, 60.0
    st = _settings(W, H, f, f, bg=(0.1, 0.2, 0.3))
    z1, z2 = 3.0, 1.5                       # index 0 is FARTHER: order must come from depth, not index
    def at(px, py, z):
        return [((2 * px + 1) / W - 1) * z * (W / (2 * f)), ((2 * py + 1) / H - 1) * z * (H / (2 * f)), z]
    out = _one([at(20, 20, z1), at(20, 20, z2)], [[0.2] * 3, [0.1] * 3], [1.0, 0.5], [[1, 0, 0], [0, 1, 0]], st)
    a_near, a_far = 0.5, 0.99               # opacity 1.0 is clamped to 0.99
    T = (1 - a_near) * (1 - a_far)
    want = torch.tensor([a_far * (1 - a_near) + T * 0.1, a_near + T * 0.2, T * 0.3], dtype=D)
    assert torch.allclose(out.color[:, 20, 20], want, atol=1e-12)
    assert abs(out.depth[0, 20, 20].item() - (a_near * z2 + (1 - a_near) * a_far * z1)) < 1e-12
    assert abs(out.opacity[0, 20, 20].item() - (1 - T)) < 1e-12
    assert out.aux["point_list"][:2].tolist() == [1, 0] or out.aux["num_rendered"] > 2


def test_near_plane_and_offscreen_culling():
    st = _settings()
    out = _one([[0, 0, 0.2], [0, 0, 0.2000001 + 1e-3], [0, 0, -1.0], [50.0, 0, 1.0]],
               [[0.05] * 3] * 4, [0.5] * 4, [[1, 1, 1]] * 4, st)
    assert out.radii.tolist()[0] == 0            # z <= 0.2 culled
    assert out.radii.tolist()[1] > 0
    assert out.radii.tolist()[2] == 0            # behind the camera
    assert out.radii.tolist()[3] == 0            # projects far outside: empty tile rectangle
    assert (out.n_touched[out.radii == 0] == 0).all()


def test_termination_and_thresholds():
    """Stack of identical opaque splats on one pixel: blending stops when T(1-a) < 1e-4 and the
    stopping splat is not blended; alpha < 1/255 contributions are skipped."""
    W, H, f = 32, 32, 40.0
    st = _settings(W, H, f, f)
    n = 12
    z = [1.0 + 0.1 * i for i in range(n)]
    means = [[((2 * 16 + 1) / W - 1) * zi * (W / (2 * f)), ((2 * 16 + 1) / H - 1) * zi * (H / (2 * f)), zi] for zi in z]
    out = _one(means, [[0.05] * 3] * n, [0.9] * n, [[1, 1, 1]] * n, st)
    # T after k splats = 0.1^k ; the 4th would leave 1e-4 * (1 - 1e-16)... : T*(1-a) = 1e-4 is NOT < 1e-4 in
    # exact arithmetic, but 0.1**4 in floating point is 1.0000000000000002e-4 -> blended; the 5th stops.
    k = int(out.aux["n_contrib"][0, 16, 16])
    T = out.aux["final_T"][0, 16, 16].item()
    assert k in (4, 5) and abs(T - 0.1 ** k) < 1e-15
    assert 0.1 ** (k + 1) < 1e-4
    tiny = _one([means[0]], [[0.05] * 3], [1.0 / 256.0], [[1, 1, 1]], st)
    assert tiny.opacity.abs().max().item() == 0 and tiny.radii[0].item() > 0 and tiny.n_touched[0].item() == 0


def test_invariants_on_random_scene():
    sc = make_scene(1500, "fr3_office", seed=4)
    st = scene_settings(sc, OracleSettings)
    out = rasterize(sc.means3D, None, sc.opacities, st, colors_precomp=sc.colors,
                    scales=sc.scales.repeat(1, 3), rotations=sc.rotations)
    assert torch.allclose(out.opacity, 1 - out.aux["final_T"], atol=1e-7)
    assert ((out.n_touched > 0) <= (out.radii > 0)).all()
    r = out.aux["ranges"]
    ne = r[r[:, 1] > r[:, 0]]
    assert ne[0, 0] == 0 and ne[-1, 1] == out.aux["num_rendered"] and (ne[1:, 0] == ne[:-1, 1]).all()
    keys = out.aux["keys"]
    assert (np.diff(keys.astype(np.int64)) >= 0).all()
    # ties (equal tile and depth bits) keep Gaussian-index order
    pl = out.aux["point_list"].numpy()
    same = np.nonzero(np.diff(keys.astype(np.int64)) == 0)[0]
    assert (pl[same + 1] > pl[same]).all()
    assert out.aux["offsets"][-1] == out.aux["num_rendered"]


def _small_problem(seed=0, n=24, W=48, H=32, sh=False, precomp=False):
    g = torch.Generator().manual_seed(seed)
    T = cam.se3_exp(torch.tensor([0.05, -0.03, 0.1, 0.04, -0.02, 0.03]))
    st = _settings(W, H, 45.0, 47.0, cx=W / 2 + 1.3, cy=H / 2 - 0.7, R=T[:3, :3], t=T[:3, 3], bg=(0.2, 0.1, 0.4))
    z = 1.0 + 3.0 * torch.rand(n, generator=g)
    pc = torch.stack([(torch.rand(n, generator=g) - 0.5) * z * 0.9, (torch.rand(n, generator=g) - 0.5) * z * 0.6, z], 1)
    means = ((pc - T[:3, 3]) @ T[:3, :3]).to(D)
    scales = (0.05 + 0.25 * torch.rand(n, 3, generator=g)).to(D)
    q = torch.randn(n, 4, generator=g)
    rot = (q / q.norm(dim=1, keepdim=True)).to(D)
    opac = (0.2 + 0.7 * torch.rand(n, 1, generator=g)).to(D)
    inp = dict(means3D=means, opacities=opac)
    if sh:
        inp["shs"] = (0.5 * torch.randn(n, 16, 3, generator=g)).to(D)
        st = st._replace(sh_degree=3)
    else:
        inp["colors_precomp"] = torch.rand(n, 3, generator=g).to(D)
    if precomp:
        inp["cov3D_precomp"] = gs_oracle.cov3d_from_scale_rot(scales, rot, torch.tensor(1.0, dtype=D))
    else:
        inp["scales"], inp["rotations"] = scales, rot
    gc = torch.randn(3, H, W, generator=g).to(D)
    gd = torch.randn(1, H, W, generator=g).to(D)
    return st, inp, gc, gd


@pytest.mark.parametrize("sh,precomp", [(False, False), (True, False), (False, True)])
def test_gradcheck_float64(sh, precomp):
    st, inp, gc, gd = _small_problem(sh=sh, precomp=precomp)
    names = list(inp.keys())

    def f(*xs):
        kw = dict(zip(names, xs))
        out = rasterize(kw["means3D"], None, kw["opacities"], st, shs=kw.get("shs"),
                        colors_precomp=kw.get("colors_precomp"), scales=kw.get("scales"),
                        rotations=kw.get("rotations"), cov3D_precomp=kw.get("cov3D_precomp"), dtype=D)
        return (out.color * gc).sum() + (out.depth * gd).sum()
    xs = [inp[k].clone().requires_grad_(True) for k in names]
    assert torch.autograd.gradcheck(f, xs, eps=1e-7, atol=1e-6, rtol=1e-4, nondet_tol=0.0)


@pytest.mark.parametrize("sh", [False, True])
def test_pose_jacobian_finite_differences(sh):
    """dL/dtheta, dL/drho from autograd at tau = 0 against central differences of the loss when the
    camera itself is moved by T_cw <- exp(tau) T_cw and every settings matrix is rebuilt."""
    st, inp, gc, gd = _small_problem(seed=3, sh=sh)
    out, g = rasterize_autograd(inp, st, gc, gd, dtype=D)
    T0 = st.viewmatrix.t().to(D)
    P = st.projmatrix_raw.t().to(D)

    def loss_at(tau):
        Tn = gs_oracle.se3_exp(tau) @ T0
        st2 = st._replace(viewmatrix=Tn.t().contiguous(), projmatrix=(P @ Tn).t().contiguous(),
                          campos=-(Tn[:3, :3].t() @ Tn[:3, 3]))
        o = rasterize(inp["means3D"], None, inp["opacities"], st2, shs=inp.get("shs"),
                      colors_precomp=inp.get("colors_precomp"), scales=inp["scales"],
                      rotations=inp["rotations"], dtype=D)
        return ((o.color * gc).sum() + (o.depth * gd).sum()).item()
    h = 1e-6
    num = []
    for i in range(6):
        e = torch.zeros(6, dtype=D)
        e[i] = h
        num.append((loss_at(e) - loss_at(-e)) / (2 * h))
    num = torch.tensor(num, dtype=D)
    ana = torch.cat([g["rho"], g["theta"]])
    assert torch.allclose(ana, num, rtol=2e-5, atol=1e-7), (ana, num)


def test_means2D_gradient_is_ndc_scaled_pixel_gradient():
    st, inp, gc, gd = _small_problem(seed=5)
    out, g = rasterize_autograd(inp, st, gc, gd, dtype=D)
    assert g["means2D"].shape == (inp["means3D"].shape[0], 3)
    assert g["means2D"][:, 2].abs().max().item() == 0
    vis = out.radii > 0
    assert g["means2D"][vis][:, :2].abs().sum().item() > 0
    assert g["means2D"][~vis].abs().sum().item() == 0


def test_argument_validation():
    st, inp, gc, gd = _small_problem()
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rasterize(inp["means3D"], None, inp["opacities"], st, scales=inp["scales"], rotations=inp["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rasterize(inp["means3D"], None, inp["opacities"], st, colors_precomp=inp["colors_precomp"])


def test_float32_and_float64_agree():
    sc = make_scene(800, "fr3_office", seed=9)
    st = scene_settings(sc, OracleSettings)
    kw = dict(colors_precomp=sc.colors, scales=sc.scales.repeat(1, 3), rotations=sc.rotations)
    a = rasterize(sc.means3D, None, sc.opacities, st, want_ambiguous=True, **kw)
    b = rasterize(sc.means3D, None, sc.opacities, st, dtype=D, **kw)
    ok = ~a.aux["ambiguous"]
    same_r = (a.radii == b.radii).float().mean().item()
    assert same_r > 0.995
    if same_r == 1.0:
        assert (a.color.double() - b.color).abs().amax(0)[ok].max() < 5e-5
