"""Fused Gaussian Adam and densification statistics vs their PyTorch definitions (pytest -m gpu)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_gaussian_adam_matches_torch_adam(native_lib):
    from monogs_amd.gaussian_optim import GaussianAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(5000, 3), (5000, 3), (5000, 1), (5000, 1), (5000, 4)]
    lrs = [1.6e-4 * 6.0, 0.0025, 0.05, 0.001, 0.001]
    a = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    fused = GaussianAdam(a, lrs)
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=0.0, eps=1e-15)
    for it in range(12):
        for pa, pb in zip(a, b):
            gr = torch.randn(*pa.shape, generator=g).to(DEV) * (10.0 ** (-it / 4))
            if it == 5:
                gr.zero_()                       # zero gradients still decay the moments and move the parameter
            pa.grad, pb.grad = gr.clone(), gr.clone()
        fused.step()
        ref.step()
        for pa, pb in zip(a, b):
            # a few float32 ulps per step (different but equivalent operation order), accumulated over 12 steps
            assert torch.allclose(pa, pb, rtol=1e-5, atol=3e-6), (it, (pa - pb).abs().max())
    assert int(fused.t_dev.item()) == 12


def test_densification_stats(native_lib):
    from monogs_amd.gaussian_optim import add_densification_stats
    g = torch.Generator().manual_seed(1)
    P = 7001
    vs = torch.randn(P, 3, generator=g).to(DEV)
    radii = torch.randint(-1, 40, (P,), generator=g).to(torch.int32).to(DEV)
    acc, den, mr = torch.rand(P, 1, generator=g).to(DEV), torch.rand(P, 1, generator=g).to(DEV), torch.rand(P, generator=g).to(DEV) * 30
    acc_r, den_r, mr_r = acc.clone(), den.clone(), mr.clone()
    vis = radii > 0
    acc_r[vis] += torch.norm(vs[vis, :2], dim=-1, keepdim=True)
    den_r[vis] += 1
    mr_r[vis] = torch.max(mr_r[vis], radii[vis].float())
    add_densification_stats(vs, radii, acc, den, mr)
    assert torch.allclose(acc, acc_r, rtol=1e-6, atol=1e-7) and torch.equal(den, den_r) and torch.equal(mr, mr_r)


@pytest.mark.parametrize("sd", [1, 3])
def test_fused_activations(native_lib, sd):
    from monogs_amd.gaussian_optim import activate
    g = torch.Generator().manual_seed(2)
    P = 4001
    raw = [torch.randn(P, 4, generator=g), torch.randn(P, sd, generator=g) - 3, torch.randn(P, 1, generator=g) * 2]
    a = [t.to(DEV).requires_grad_(True) for t in raw]
    b = [t.to(DEV).requires_grad_(True) for t in raw]
    w = [torch.randn(P, 4, generator=g).to(DEV), torch.randn(P, 3, generator=g).to(DEV), torch.randn(P, 1, generator=g).to(DEV)]
    outs = activate(*a)
    sc = torch.exp(b[1])
    refs = (torch.nn.functional.normalize(b[0]), sc.repeat(1, 3) if sd == 1 else sc, torch.sigmoid(b[2]))
    for o, r in zip(outs, refs):
        assert torch.allclose(o, r, rtol=1e-6, atol=1e-7)
    sum((o * ww).sum() for o, ww in zip(outs, w)).backward()
    sum((r * ww).sum() for r, ww in zip(refs, w)).backward()
    for x, y in zip(a, b):
        assert torch.allclose(x.grad, y.grad, rtol=1e-5, atol=1e-6), (x.grad - y.grad).abs().max()
