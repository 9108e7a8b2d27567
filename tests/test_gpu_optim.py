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
                gr.zero_()                       # explicit zero gradients still decay the moments and move the parameter
            pa.grad, pb.grad = gr.clone(), gr.clone()
        fused.step()
        ref.step()
        for pa, pb in zip(a, b):
            # a few float32 ulps per step (different but equivalent operation order), accumulated over 12 steps
            assert torch.allclose(pa, pb, rtol=1e-5, atol=3e-6), (it, (pa - pb).abs().max())
    assert fused.t_dev.tolist() == [12] * 5


def _reference_surgery(opt, names):
    """The reference's optimiser-state surgery, restated on torch.optim.Adam for the check below
    (/root/reference/gaussian_splatting/scene/gaussian_model.py:642-743)."""
    def group(name):
        return next(g for g in opt.param_groups if g["name"] == name)

    def cat(ext):                                   # cat_tensors_to_optimizer
        for n in names:
            g = group(n)
            st = opt.state.get(g["params"][0])
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext[n])), 0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext[n])), 0)
            del opt.state[g["params"][0]]
            g["params"][0] = torch.nn.Parameter(torch.cat((g["params"][0], ext[n]), 0).requires_grad_(True))
            opt.state[g["params"][0]] = st

    def prune(mask):                                # _prune_optimizer
        for n in names:
            g = group(n)
            st = opt.state.get(g["params"][0])
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][mask], st["exp_avg_sq"][mask]
            del opt.state[g["params"][0]]
            g["params"][0] = torch.nn.Parameter(g["params"][0][mask].requires_grad_(True))
            opt.state[g["params"][0]] = st

    def replace(name, tensor):                      # replace_tensor_to_optimizer
        g = group(name)
        st = opt.state.get(g["params"][0])
        st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(tensor), torch.zeros_like(tensor)
        del opt.state[g["params"][0]]
        g["params"][0] = torch.nn.Parameter(tensor.requires_grad_(True))
        opt.state[g["params"][0]] = st
    return cat, prune, replace


def test_adam_state_surgery_matches_torch(native_lib):
    """extend / prune / replace carry the moments and step counts exactly as the reference's surgery on
    torch.optim.Adam does: same parameters after steps interleaved with a cat, a prune and an opacity-style replace
    (the replaced tensor has no gradient on the step that follows -- torch skips it, and so does the fused step)."""
    from monogs_amd.gaussian_optim import GaussianAdam
    names = ["xyz", "f_dc", "opacity", "scaling", "rotation"]
    widths = [3, 3, 1, 1, 4]
    lrs = [1.6e-4 * 6.0, 0.0025, 0.05, 0.001, 0.001]
    g = torch.Generator().manual_seed(3)
    rnd = lambda n, w: torch.randn(n, w, generator=g).to(DEV)  # noqa: E731
    P = 3000
    a = [rnd(P, w).requires_grad_(True) for w in widths]
    fused = GaussianAdam(a, lrs)
    ref = torch.optim.Adam([{"params": [torch.nn.Parameter(t.detach().clone())], "lr": lr, "name": n}
                            for t, lr, n in zip(a, lrs, names)], lr=0.0, eps=1e-15)
    cat, prune, replace = _reference_surgery(ref, names)

    def ref_params():
        return [next(gr for gr in ref.param_groups if gr["name"] == n)["params"][0] for n in names]

    def step(skip=()):
        for i, (pa, pb) in enumerate(zip(fused.params, ref_params())):
            if i in skip:
                pa.grad = pb.grad = None
                continue
            gr = torch.randn(*pa.shape, generator=g).to(DEV)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        fused.step()
        ref.step()

    def check(tag):
        for n, pa, pb in zip(names, fused.params, ref_params()):
            assert pa.shape == pb.shape, (tag, n)
            assert torch.allclose(pa, pb, rtol=1e-5, atol=3e-6), (tag, n, (pa - pb).abs().max())
        for i, n in enumerate(names):
            st = ref.state[ref_params()[i]]
            assert torch.allclose(fused.exp_avg[i], st["exp_avg"], rtol=1e-5, atol=1e-7), (tag, n)
            assert torch.allclose(fused.exp_avg_sq[i], st["exp_avg_sq"], rtol=1e-5, atol=1e-9), (tag, n)
            assert int(fused.t_dev[i]) == int(st["step"]), (tag, n)

    for _ in range(3):
        step()
    check("start")
    ext = {n: rnd(500, w) for n, w in zip(names, widths)}           # densification_postfix
    fused.extend([ext[n] for n in names]); cat(ext)
    assert fused.params[0].shape[0] == P + 500 and fused.params[0].requires_grad and fused.params[0].is_leaf
    step(); step()
    check("after extend")
    mask = (torch.rand(P + 500, generator=g) > 0.3).to(DEV)        # prune_points
    fused.prune(mask); prune(mask)
    step()
    check("after prune")
    new_op = rnd(int(mask.sum()), 1)                                # reset_opacity*
    fused.replace(2, new_op); replace("opacity", new_op.clone())
    step(skip=(2,))          # the mapper steps right after the reset: the new opacity tensor has no gradient yet
    check("after replace (opacity skipped)")
    assert int(fused.t_dev[2]) == int(fused.t_dev[0]) - 1
    step(); step()
    check("end")
    fused.zero_grad()
    fused.step()             # no gradients at all: nothing moves, no count advances
    check("all grads None")


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


def test_fan_out_sums_the_consumers_gradients_in_one_launch(native_lib):
    """fan_out(n, *tensors): n consumers of the same tensors; the gradients must equal plain autograd's pairwise sums --
    through the flat path (each consumer returns its gradients carved from one buffer, as the rasteriser does) and
    through the per-tensor path (separate buffers, some of them missing)."""
    from monogs_amd.gaussian_optim import fan_out, sum_buffers
    g = torch.Generator().manual_seed(3)
    shapes = [(1000, 3), (1000, 3), (1000, 1), (1000, 3), (1000, 4)]
    leaves = [torch.randn(s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    n = 5
    w = [[torch.randn(s, generator=g).to(DEV) for s in shapes] for _ in range(n)]

    class Carved(torch.autograd.Function):          # a consumer whose backward returns views of ONE allocation
        @staticmethod
        def forward(ctx, k, *ts):
            ctx.k = k
            return sum((t * w[k][j]).sum() for j, t in enumerate(ts))

        @staticmethod
        def backward(ctx, go):
            flat = torch.cat([(w[ctx.k][j] * go).reshape(-1) for j in range(len(shapes))])
            out, o = [], 0
            for s in shapes:
                m = s[0] * s[1]
                out.append(flat[o:o + m].view(s))
                o += m
            return (None, *out)

    def ref():
        for t in leaves:
            t.grad = None
        sum(sum((t * w[k][j]).sum() for j, t in enumerate(leaves)) for k in range(n)).backward()
        return [t.grad.clone() for t in leaves]
    want = ref()
    for mode in ("flat", "separate"):
        for t in leaves:
            t.grad = None
        fans = fan_out(n, *leaves)
        if mode == "flat":
            total = sum(Carved.apply(k, *fans[k]) for k in range(n))
        else:       # consumer 0 never touches tensor 2: its gradient arrives as None
            total = sum(sum((t * w[k][j]).sum() for j, t in enumerate(fans[k]) if not (k == 0 and j == 2)) for k in range(n))
        total.backward()
        for j, (t, r) in enumerate(zip(leaves, want)):
            r = r - w[0][2] if (mode == "separate" and j == 2) else r
            assert torch.allclose(t.grad, r, rtol=1e-5, atol=1e-5), (mode, j)
    many = [torch.randn(777, generator=g).to(DEV) for _ in range(37)]       # more than 16 sources: groups
    assert torch.allclose(sum_buffers(many), torch.stack(many).sum(0), rtol=1e-5, atol=1e-5)
