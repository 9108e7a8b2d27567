"""Host-side surface of the drop-in (no GPU): names, settings tuple, argument errors."""
import inspect

import pytest
import torch


def test_dropin_module_names_and_symbols():
    import diff_gaussian_rasterization as dgr
    from simple_knn._C import distCUDA2
    assert callable(distCUDA2)
    fields = dgr.GaussianRasterizationSettings._fields
    assert fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                      "projmatrix", "projmatrix_raw", "sh_degree", "campos", "prefiltered", "debug")
    sig = inspect.signature(dgr.GaussianRasterizer.forward)
    assert list(sig.parameters)[1:] == ["means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                        "rotations", "cov3D_precomp", "theta", "rho"]
    assert hasattr(dgr.GaussianRasterizer, "markVisible")


def _rs():
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    e = torch.eye(4)
    return GaussianRasterizationSettings(48, 64, 0.5, 0.4, torch.zeros(3), 1.0, e, e, e, 0, torch.zeros(3), False, False)


def test_exactly_one_of_errors_match_upstream_messages():
    from diff_gaussian_rasterization import GaussianRasterizer
    r = GaussianRasterizer(_rs())
    m, o = torch.zeros(4, 3), torch.zeros(4, 1)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=o, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=o, shs=torch.zeros(4, 1, 3), colors_precomp=m, scales=m,
          rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=o, colors_precomp=m)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=o, colors_precomp=m, scales=m, rotations=torch.zeros(4, 4),
          cov3D_precomp=torch.zeros(4, 6))


def test_cpu_tensors_fail_loudly(native_lib):
    from diff_gaussian_rasterization import GaussianRasterizer
    r = GaussianRasterizer(_rs())
    m, o = torch.zeros(4, 3), torch.zeros(4, 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=m, means2D=m, opacities=o, colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))


def test_render_returns_none_for_empty_map():
    from monogs_amd.renderer import render
    assert render(None, None, torch.zeros(0, 3), None, None, None, None, None) is None


def test_synthetic_scene_is_deterministic():
    from monogs_amd.synthetic import make_scene
    a, b = make_scene(500, "fr3_office", seed=3), make_scene(500, "fr3_office", seed=3)
    for x, y in zip(a, b):
        if isinstance(x, torch.Tensor):
            assert torch.equal(x, y)
    c = make_scene(500, "fr3_office", seed=4)
    assert not torch.equal(a.means3D, c.means3D)


def test_reset_opacity_nonvisible_as_the_reference():
    """/root/reference/gaussian_splatting/scene/gaussian_model.py:527-535, restated line by line: the new raw opacity is
    inverse_sigmoid(0.4) everywhere, then ``opacities_new[filter] = self.get_opacity[filter]`` for every visibility
    filter -- the ACTIVATED value goes into the raw parameter -- and the Adam moments of the tensor restart from zero
    (``replace_tensor_to_optimizer``, :642-656).  ``as_reference=False`` keeps the raw value of the visible ones."""
    from monogs_amd.gaussian_map import GaussianMap, inverse_sigmoid
    g = torch.Generator().manual_seed(5)
    P = 40
    raw = torch.randn(P, 1, generator=g)
    filters = [torch.rand(P, generator=g) > 0.6, torch.rand(P, generator=g) > 0.7]
    for as_ref in (True, False):
        m = GaussianMap("cpu")
        m.densification_postfix(torch.randn(P, 3, generator=g), torch.rand(P, 3, generator=g), raw.clone(),
                                torch.randn(P, 1, generator=g), torch.randn(P, 4, generator=g))
        m.optimizer.exp_avg[2].fill_(0.3)
        m.optimizer.exp_avg_sq[2].fill_(0.2)
        expected = inverse_sigmoid(torch.ones(P, 1) * 0.4)
        for f in filters:
            expected[f] = (torch.sigmoid(raw) if as_ref else raw)[f]
        m.reset_opacity_nonvisible(filters, as_reference=as_ref)
        assert torch.equal(m._opacity.detach(), expected) and m._opacity.requires_grad
        assert m.optimizer.params[2] is m._opacity
        assert float(m.optimizer.exp_avg[2].abs().max()) == 0.0 and float(m.optimizer.exp_avg_sq[2].abs().max()) == 0.0
