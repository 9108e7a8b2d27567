"""The oracle's per-Gaussian projection and per-fragment alpha rule against the one piece of splatting code the reference
DOES hold: its OpenGL viewer's shaders (the CUDA rasteriser is an un-vendored submodule).

`/root/reference/viewer/gl_render/shaders/gau_vert.glsl:60-107,149-154` (computeCov3D, computeCov2D with the 1.3 x tan(fov)
clamp and the +0.3 low-pass, the conic) and `gau_frag.glsl:20-25` (power, `power > 0` discard, `min(0.99, alpha * exp(power))`,
the 1/255 discard) are restated below in numpy with GLSL's semantics (column-major `mat3(...)` constructors, `M[i]` = column
i) and compared with `oracle.gs_oracle.preprocess` / `rasterize` on random inputs.  This pins those formulas of the oracle --
hence of the HIP kernels, which are checked against the oracle -- to the reference authors' own implementation of the same
algorithm; the tile binning and the blending ORDER have no counterpart in the shaders and stay pinned by construction only.
CPU only; needs nothing from /root/reference at run time (the shader text is restated, not read)."""
import numpy as np
import torch

from monogs_amd.synthetic import make_scene, scene_settings
from oracle import OracleSettings, gs_oracle


# ---- GLSL helpers: matrices are kept in the mathematical [row, col] convention ----------------------------------------------
def mat3(*a):
    """GLSL mat3(a0..a8): column-major -- a0..a2 is the first COLUMN."""
    return np.array(a, dtype=np.float64).reshape(3, 3).T


def compute_cov3d(scale, q):
    """gau_vert.glsl:60-80"""
    S = np.diag(scale.astype(np.float64))
    r, x, y, z = q
    R = mat3(1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y - r * z), 2.0 * (x * z + r * y),
             2.0 * (x * y + r * z), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z - r * x),
             2.0 * (x * z - r * y), 2.0 * (y * z + r * x), 1.0 - 2.0 * (x * x + y * y))
    M = S @ R
    return M.T @ M


def compute_cov2d(mean_view, focal_x, focal_y, tan_fovx, tan_fovy, cov3d, view3):
    """gau_vert.glsl:82-107; `view3` = mat3(viewmatrix), the rotation block of the world->view matrix."""
    t = mean_view.astype(np.float64).copy()
    limx, limy = 1.3 * tan_fovx, 1.3 * tan_fovy
    txtz, tytz = t[0] / t[2], t[1] / t[2]
    t[0] = min(limx, max(-limx, txtz)) * t[2]
    t[1] = min(limy, max(-limy, tytz)) * t[2]
    J = mat3(focal_x / t[2], 0.0, -(focal_x * t[0]) / (t[2] * t[2]),
             0.0, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2]),
             0.0, 0.0, 0.0)
    W = view3.T
    T = W @ J
    cov = T.T @ cov3d.T @ T
    cov[0][0] += 0.3
    cov[1][1] += 0.3
    # GLSL cov[0][1] = column 0, row 1; the matrix is symmetric
    return np.array([cov[0][0], cov[1][0], cov[1][1]])


def conic_of(cov2d):
    """gau_vert.glsl:149-154"""
    det = cov2d[0] * cov2d[2] - cov2d[1] * cov2d[1]
    det_inv = 1.0 / det
    return np.array([cov2d[2] * det_inv, -cov2d[1] * det_inv, cov2d[0] * det_inv])


def fragment_alpha(conic, alpha, coordxy):
    """gau_frag.glsl:20-25; a discarded fragment contributes nothing."""
    power = -0.5 * (conic[0] * coordxy[0] * coordxy[0] + conic[2] * coordxy[1] * coordxy[1]) - conic[1] * coordxy[0] * coordxy[1]
    if power > 0.0:
        return 0.0
    opacity = min(0.99, alpha * np.exp(power))
    if opacity < 1.0 / 255.0:
        return 0.0
    return opacity


def _scene(n, seed, anisotropic=True):
    sc = make_scene(n, "fr3_office", seed=seed, anisotropic=anisotropic)
    return sc, scene_settings(sc, OracleSettings)


def test_projection_matches_the_viewer_shader():
    sc, st = _scene(400, seed=31)
    g = gs_oracle.preprocess(sc.means3D.double(), sc.scales.double(), sc.rotations.double(), sc.opacities.double(), st,
                             colors_precomp=sc.colors.double(), dtype=torch.float64)
    V = st.viewmatrix.double().numpy().T            # the rasteriser takes the transposed (row-vector) matrix
    W_, H_ = int(st.image_width), int(st.image_height)
    fx, fy = W_ / (2.0 * st.tanfovx), H_ / (2.0 * st.tanfovy)
    rot = sc.rotations.double().numpy()
    n_checked = n_clamped = 0
    for i in range(sc.means3D.shape[0]):
        if not bool(g["visible"][i]):
            continue
        p = np.append(sc.means3D[i].double().numpy(), 1.0)
        mean_view = V @ p
        s3 = sc.scales[i].double().numpy() * float(st.scale_modifier)
        cov3d = compute_cov3d(s3 if s3.size == 3 else np.repeat(s3, 3), rot[i])
        o6 = g["cov3D"][i].numpy()
        want3 = np.array([cov3d[0, 0], cov3d[0, 1], cov3d[0, 2], cov3d[1, 1], cov3d[1, 2], cov3d[2, 2]])
        assert np.allclose(o6, want3, rtol=1e-12, atol=1e-14)
        cov2d = compute_cov2d(mean_view[:3], fx, fy, float(st.tanfovx), float(st.tanfovy), cov3d, V[:3, :3])
        assert np.allclose(g["cov2D"][i].numpy(), cov2d, rtol=1e-10, atol=1e-12)
        assert np.allclose(g["conic"][i].numpy(), conic_of(cov2d), rtol=1e-9, atol=1e-12)
        n_checked += 1
        n_clamped += int(abs(mean_view[0] / mean_view[2]) > 1.3 * st.tanfovx or abs(mean_view[1] / mean_view[2]) > 1.3 * st.tanfovy)
    assert n_checked > 100


def test_fov_clamp_branch_matches_the_viewer_shader():
    """Gaussians far outside the frustum (still in front of the camera) take the 1.3 x tan(fov) clamp of t.x / t.y."""
    sc, st = _scene(50, seed=32)
    V = st.viewmatrix.double().numpy().T
    R, t = V[:3, :3], V[:3, 3]
    g_ = torch.Generator().manual_seed(5)
    view_pts = torch.stack([torch.empty(50).uniform_(-8, 8, generator=g_), torch.empty(50).uniform_(-6, 6, generator=g_),
                            torch.empty(50).uniform_(1.0, 3.0, generator=g_)], 1).double().numpy()
    world = (R.T @ (view_pts - t).T).T
    g = gs_oracle.preprocess(torch.from_numpy(world), sc.scales.double(), sc.rotations.double(), sc.opacities.double(), st,
                             colors_precomp=sc.colors.double(), dtype=torch.float64)
    W_, H_ = int(st.image_width), int(st.image_height)
    fx, fy = W_ / (2.0 * st.tanfovx), H_ / (2.0 * st.tanfovy)
    clamped = 0
    for i in range(50):
        mv = V @ np.append(world[i], 1.0)
        s3 = sc.scales[i].double().numpy() * float(st.scale_modifier)
        cov3d = compute_cov3d(s3 if s3.size == 3 else np.repeat(s3, 3), sc.rotations[i].double().numpy())
        cov2d = compute_cov2d(mv[:3], fx, fy, float(st.tanfovx), float(st.tanfovy), cov3d, R)
        assert np.allclose(g["cov2D"][i].numpy(), cov2d, rtol=1e-10, atol=1e-12)
        clamped += int(abs(mv[0] / mv[2]) > 1.3 * st.tanfovx or abs(mv[1] / mv[2]) > 1.3 * st.tanfovy)
    assert clamped >= 10


def test_fragment_rule_matches_the_viewer_shader():
    """One Gaussian over a black background: colour = c * alpha(pixel), opacity image = alpha, with alpha the fragment
    shader's (power > 0 and alpha < 1/255 discarded, 0.99 cap)."""
    sc, st = _scene(60, seed=33)
    H_, W_ = int(st.image_height), int(st.image_width)
    g = gs_oracle.preprocess(sc.means3D.double(), sc.scales.double(), sc.rotations.double(), sc.opacities.double(), st,
                             colors_precomp=sc.colors.double(), dtype=torch.float64)
    vis = torch.nonzero(g["visible"]).reshape(-1)
    inside = [int(i) for i in vis if 20 < float(g["xy"][i, 0]) < W_ - 20 and 20 < float(g["xy"][i, 1]) < H_ - 20]
    checked = capped = cut = 0
    for i in inside[:6]:
        big = checked % 2 == 0        # every other one: four times the size and opacity 0.999, so that the 0.99 cap is reached
        opac = torch.tensor([0.999 if big else float(sc.opacities[i])], dtype=torch.float64)
        scales = sc.scales[i:i + 1].double() * (4.0 if big else 1.0)
        gi = gs_oracle.preprocess(sc.means3D[i:i + 1].double(), scales, sc.rotations[i:i + 1].double(), opac, st,
                                  colors_precomp=sc.colors[i:i + 1].double(), dtype=torch.float64)
        out = gs_oracle.rasterize(sc.means3D[i:i + 1].double(), torch.zeros(1, 3, dtype=torch.float64), opac, st,
                                  colors_precomp=sc.colors[i:i + 1].double(), scales=scales,
                                  rotations=sc.rotations[i:i + 1].double(), dtype=torch.float64)
        conic, (cx, cy) = gi["conic"][0].numpy(), gi["xy"][0].numpy()
        r = int(gi["radii"][0])
        ys = sorted(set(range(max(0, int(cy) - r - 2), min(H_, int(cy) + r + 3), 3)) | {int(round(cy))})
        xs = sorted(set(range(max(0, int(cx) - r - 2), min(W_, int(cx) + r + 3), 3)) | {int(round(cx))})
        for py in ys:
            for px in xs:
                a = fragment_alpha(conic, float(opac), (cx - px, cy - py))
                assert abs(float(out.opacity[0, py, px]) - a) < 1e-9
                assert np.allclose(out.color[:, py, px].numpy(), sc.colors[i].double().numpy() * a, atol=1e-9)
                capped += int(a == 0.99)
                cut += int(a == 0.0)
        checked += 1
    assert checked >= 4 and capped > 0 and cut > 0


def test_sh_constants_match_the_viewer_shader():
    """gau_vert.glsl:3-17 (the same constants as gaussian_splatting/utils/sh_utils.py)."""
    assert gs_oracle.SH_C0 == 0.28209479177387814 and gs_oracle.SH_C1 == 0.4886025119029199
    assert list(gs_oracle.SH_C2) == [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
                                     0.5462742152960396]
    assert list(gs_oracle.SH_C3) == [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                                     -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def test_cov3d_matches_the_reference_python_covariance():
    """`GaussianModel.build_covariance_from_scaling_rotation` (/root/reference/gaussian_splatting/scene/gaussian_model.py:76-82)
    with `build_scaling_rotation` / `build_rotation` / `strip_symmetric` (gaussian_splatting/utils/general_utils.py:97-148),
    restated in numpy (they allocate on "cuda" and cannot run in the build container): L = R diag(mod * s), Sigma = L L^T, six
    upper-triangular entries in the order (xx, xy, xz, yy, yz, zz).  The rasteriser receives unit quaternions (the rotation
    activation is `normalize`, gaussian_model.py:68), for which `build_rotation`'s own normalisation is the identity."""
    sc, st = _scene(300, seed=34)
    q = sc.rotations.double().numpy()
    q = q / np.sqrt((q * q).sum(1, keepdims=True))                                   # build_rotation:114-118
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.zeros((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z); R[:, 0, 2] = 2 * (x * z + r * y)   # noqa: E702
    R[:, 1, 0] = 2 * (x * y + r * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)   # noqa: E702
    R[:, 2, 0] = 2 * (x * z - r * y); R[:, 2, 1] = 2 * (y * z + r * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)   # noqa: E702
    s = sc.scales.double().numpy() * 1.7
    s = s if s.shape[1] == 3 else np.repeat(s, 3, axis=1)
    L = R @ np.stack([np.diag(v) for v in s])                                        # build_scaling_rotation:139-148
    cov = L @ L.transpose(0, 2, 1)
    want = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)   # strip_lowerdiag
    got = gs_oracle.cov3d_from_scale_rot(torch.from_numpy(s / 1.7), torch.from_numpy(q), torch.tensor(1.7, dtype=torch.float64)).numpy()
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14)
