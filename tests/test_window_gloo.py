"""Mapping-window sharding over 2 ranks on CPU (gloo).

* bucket mechanics on seeded random gradient tensors (both bucket modes);
* the exchanges of SURVEY.md section 8e with the ORACLE as the per-keyframe renderer (the product has no CPU renderer):
  the all-reduced Gaussian gradients equal the single-process backward of the summed loss, the per-keyframe
  visibility bits / statistics / poses arrive on every rank.  The same window with the HIP rasteriser runs in
  tests/test_gpu_window.py."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _keyframe_grads(P, k):
    """Deterministic stand-in for one keyframe's backward: gradients of a fixed loss."""
    g = torch.Generator().manual_seed(100 + k)
    return [torch.randn(P, w, generator=g) for w in (3, 3, 1, 1, 4)], torch.rand(P, generator=g), \
        (torch.rand(P, generator=g) > 0.5).float(), torch.randint(0, 30, (P,), generator=g).float()


def _worker(rank, world, port, P, n_kf, out, per_tensor):
    sys.path.insert(0, ROOT)
    from monogs_amd.window import allreduce_window_grads, shard_keyframes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = [torch.zeros(P, w, requires_grad=True) for w in (3, 3, 1, 1, 4)]
    for p in params:
        p.grad = torch.zeros_like(p)
    norm = torch.zeros(P)
    vis = torch.zeros(P)
    maxr = torch.zeros(P)
    for k in shard_keyframes(n_kf, rank, world):
        gs, n, v, r = _keyframe_grads(P, k)
        for p, g in zip(params, gs):
            p.grad += g
        norm += n
        vis += v
        maxr = torch.maximum(maxr, r)
    from monogs_amd.window import GradBucket
    bucket = GradBucket(params, extra_cols=2, per_tensor=per_tensor)
    _, norm_s, vis_s, maxr_s = allreduce_window_grads(params, norm, vis, maxr, bucket=bucket)
    if rank == 0:
        torch.save(dict(grads=[p.grad for p in params], norm=norm_s.clone(), vis=vis_s.clone(), maxr=maxr_s), out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_keyframes_partition():
    from monogs_amd.window import shard_keyframes
    for world in (1, 2, 3, 8):
        owned = [shard_keyframes(10, r, world) for r in range(world)]
        assert sorted(k for o in owned for k in o) == list(range(10))
    assert shard_keyframes(8, 3, 8) == [3]


@pytest.mark.parametrize("per_tensor", [False, True])
def test_two_rank_allreduce_equals_serial_sum(tmp_path, per_tensor):
    P, n_kf, world = 257, 5, 2
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, _free_port(), P, n_kf, out, per_tensor), nprocs=world, join=True)
    got = torch.load(out)
    want = [torch.zeros(P, w) for w in (3, 3, 1, 1, 4)]
    norm, vis, maxr = torch.zeros(P), torch.zeros(P), torch.zeros(P)
    for k in range(n_kf):
        gs, n, v, r = _keyframe_grads(P, k)
        for w, g in zip(want, gs):
            w += g
        norm += n
        vis += v
        maxr = torch.maximum(maxr, r)
    for a, b in zip(got["grads"], want):
        assert torch.allclose(a, b, atol=1e-6)
    assert torch.allclose(got["norm"], norm, atol=1e-6)
    assert torch.allclose(got["vis"], vis)
    assert torch.equal(got["maxr"], maxr)


def test_single_process_bucket_roundtrip():
    from monogs_amd.window import GradBucket
    params = [torch.zeros(9, w, requires_grad=True) for w in (3, 1, 4)]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(i + 1))
    b = GradBucket(params)
    b.pack()
    assert b.buf.shape == (9, 8)
    for p in params:
        p.grad.zero_()
    b.unpack()
    assert [float(p.grad[0, 0]) for p in params] == [1.0, 2.0, 3.0]


# ---- the section-8e exchanges with real (oracle) renders ------------------------------------------------------------
class _KF:
    """Minimal keyframe: pose + exposure, as window.all_gather_poses reads and writes them."""

    def __init__(self, k):
        from oracle import gs_oracle
        T = gs_oracle.se3_exp(torch.tensor([0.05 * k, -0.03 * k, 0.02 * k, 0.01 * k, 0.02 * k, 0.0], dtype=torch.float64)).float()
        self.R, self.T = T[:3, :3].contiguous(), T[:3, 3].contiguous()
        self.exposure_a = torch.nn.Parameter(torch.tensor([0.01 * k]))
        self.exposure_b = torch.nn.Parameter(torch.tensor([-0.02 * k]))


def _oracle_keyframe(params, k):
    """Render keyframe k of a tiny scene with the oracle and return (loss, viewspace tensor, radii, n_touched)."""
    from monogs_amd.synthetic import make_scene, scene_settings
    from oracle import OracleSettings, gs_oracle, rasterize
    intr = dict(fx=60.0, fy=60.0, cx=32.0, cy=24.0, W=64, H=48)
    sc = make_scene(params[0].shape[0], intr, seed=5, mean_radius_px=3.0,
                    pose_tau=(0.1 + 0.02 * k, -0.2, 0.3 - 0.03 * k, 0.05, 0.02 * k, -0.04))
    st = scene_settings(sc, OracleSettings)
    xyz, rgb, opac, sca, rot = params
    m2d = torch.zeros_like(xyz, requires_grad=True)
    out = rasterize(xyz, m2d, torch.sigmoid(opac), st, colors_precomp=rgb, scales=torch.exp(sca).repeat(1, 3),
                    rotations=torch.nn.functional.normalize(rot), dtype=torch.float32)
    loss = (out.color * sc.grad_color).sum() * 1e3 + (out.depth * sc.grad_depth).sum() * 1e3
    return loss, m2d, out.radii, out.n_touched


def _scene_params(P):
    from monogs_amd.synthetic import make_scene
    intr = dict(fx=60.0, fy=60.0, cx=32.0, cy=24.0, W=64, H=48)
    sc = make_scene(P, intr, seed=5, mean_radius_px=3.0)
    return [sc.means3D.clone().requires_grad_(True), sc.colors.clone().requires_grad_(True),
            torch.logit(sc.opacities.clamp(0.05, 0.95)).requires_grad_(True), torch.log(sc.scales).requires_grad_(True),
            sc.rotations.clone().requires_grad_(True)]


def _window_local(params, keyframes):
    """What one rank does for its keyframes: summed loss, one backward, per-keyframe statistics."""
    P = params[0].shape[0]
    loss, per_kf = None, {}
    for k in keyframes:
        l, m2d, radii, n_touched = _oracle_keyframe(params, k)
        loss = l if loss is None else loss + l
        per_kf[k] = (m2d, radii, n_touched)
    if loss is not None:
        loss.backward()
    norm, vis, maxr = torch.zeros(P), torch.zeros(P), torch.zeros(P)
    for k, (m2d, radii, _) in per_kf.items():
        v = radii > 0
        norm[v] += m2d.grad[v, :2].norm(dim=-1)
        vis[v] += 1
        maxr[v] = torch.maximum(maxr[v], radii[v].float())
    return norm, vis, maxr, {k: t[2] for k, t in per_kf.items()}


def _exchange_worker(rank, world, port, P, n_kf, out):
    sys.path.insert(0, ROOT)
    from monogs_amd import window as W
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    params = _scene_params(P)
    mine = W.shard_keyframes(n_kf, rank, world)
    norm, vis, maxr, touched = _window_local(params, mine)
    _, norm_s, vis_s, maxr_s = W.allreduce_window_grads(params, norm, vis, maxr)
    vis_bits = W.all_gather_visibility(touched, n_kf, P)
    kfs = [_KF(k) if k % world == rank else _KF(0) for k in range(n_kf)]      # non-owned copies are stale
    W.all_gather_poses(kfs)
    g1, g2 = W.split_generator("cpu", 7, 150), W.split_generator("cpu", 7, 151)
    torch.save(dict(grads=[p.grad.clone() for p in params], norm=norm_s.clone(), vis=vis_s.clone(), maxr=maxr_s.clone(),
                    bits=vis_bits, poses=[(k.R, k.T, k.exposure_a.data, k.exposure_b.data) for k in kfs],
                    draws=(torch.randn(4, generator=g1), torch.randn(4, generator=g2)),
                    sync=W.replicas_in_sync(params), sync_bad=W.replicas_in_sync([torch.full((3,), float(rank))])),
               f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchanges_with_oracle_renders(tmp_path):
    P, n_kf, world = 120, 3, 2
    out = str(tmp_path / "x")
    mp.spawn(_exchange_worker, args=(world, _free_port(), P, n_kf, out), nprocs=world, join=True)
    r0, r1 = torch.load(f"{out}.0"), torch.load(f"{out}.1")
    params = _scene_params(P)
    norm, vis, maxr, touched = _window_local(params, range(n_kf))         # the single-process window
    for a, b, p in zip(r0["grads"], r1["grads"], params):
        assert torch.equal(a, b)                                           # every rank holds the same reduced bits
        # (isotropic scales: the rotation gradient is identically zero up to rounding noise, hence the absolute term)
        assert (a - p.grad).norm() <= 1e-5 * p.grad.norm() + 1e-6
    assert all(p.grad.abs().max() > 1e-4 for p in params[:4])
    assert torch.allclose(r0["norm"], norm, rtol=1e-5, atol=1e-9) and torch.equal(r0["vis"], vis)
    assert torch.equal(r0["maxr"], maxr) and torch.equal(r1["maxr"], maxr)
    for k in range(n_kf):
        want = touched[k] > 0
        assert want.any() and torch.equal(r0["bits"][k], want) and torch.equal(r1["bits"][k], want)
    for k in range(n_kf):
        ref = _KF(k)
        for r in (r0, r1):
            got = r["poses"][k]
            assert torch.equal(got[0], ref.R) and torch.equal(got[1], ref.T)
            assert torch.equal(got[2], ref.exposure_a.data) and torch.equal(got[3], ref.exposure_b.data)
    assert torch.equal(r0["draws"][0], r1["draws"][0]) and not torch.equal(r0["draws"][0], r0["draws"][1])
    assert r0["sync"] and r1["sync"] and not r0["sync_bad"] and not r1["sync_bad"]


def test_flat_view_finds_back_to_back_gradients():
    from monogs_amd.window import flat_view
    P = 37
    flat = torch.arange(P * 12, dtype=torch.float32)
    views, o = [], 0
    for w in (3, 3, 1, 1, 4):
        views.append(flat[o:o + P * w].view(P, w))
        o += P * w
    fv = flat_view(views)
    assert fv is not None and fv.shape == (P * 12,) and fv.data_ptr() == flat.data_ptr()
    fv.mul_(2)                                              # aliases the gradients
    assert torch.equal(views[4], (flat[P * 8:]).view(P, 4)) and float(views[0][0, 1]) == 2.0
    assert flat_view([views[0], views[2]]) is None         # a gap
    assert flat_view([views[0], torch.zeros(P, 3)]) is None  # another storage
    assert flat_view([views[0].t()]) is None               # not contiguous
    assert flat_view([]) is None


def _pipeline_worker(rank, world, port, P, n_kf, out):
    """The per-keyframe exchange schedule on deterministic stand-in gradients: overlapped (collective j issued right behind
    producer j) against the same collectives issued after the last producer, and against the one-bucket exchange."""
    sys.path.insert(0, ROOT)
    from monogs_amd.window import all_reduce_, pipelined_all_reduce, rows_per_rank, shard_keyframes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_keyframes(n_kf, rank, world)
    rows = rows_per_rank(n_kf, world)         # the SAME number of slot collectives on every rank, whatever it owns
    res = {}
    for overlap in (True, False):
        bufs = [torch.full((P * 14,), 7.0) for _ in range(rows)]       # (stale contents: an unowned slot must be cleared)
        order = []

        def produce(j):
            order.append(j)
            gs = _keyframe_grads(P, mine[j])[0]
            bufs[j].copy_(torch.cat([g.t().reshape(-1) for g in gs] + [torch.zeros(P * 2)]))      # [cols][P], as the bucket
        pipelined_all_reduce(rows, produce, bufs, overlap=overlap, n_owned=len(mine))
        total = bufs[0].clone()
        for b in bufs[1:]:               # fixed order: owned keyframe 0, 1, ...
            total += b
        res[overlap] = (total, list(order))
    bucket = torch.zeros(P * 14)
    for k in mine:                       # the one-bucket exchange: local sum first, one all-reduce
        gs = _keyframe_grads(P, k)[0]
        bucket += torch.cat([g.t().reshape(-1) for g in gs] + [torch.zeros(P * 2)])
    all_reduce_(bucket)
    torch.save(dict(overlap=res[True][0], serial=res[False][0], order=res[True][1], bucket=bucket), f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_kf", [6, 3, 5])
def test_pipelined_exchange_is_bit_identical_to_the_unpipelined_schedule(tmp_path, n_kf):
    """`WindowMapper.exchange = "per_keyframe"` (window / world > 1): keyframe k's bucket is all-reduced while keyframe k + 1
    renders.  Issuing the collectives early must change nothing: overlapped == issued-after-the-last-producer, bit for bit, on
    both ranks; both ranks hold the same bits; and the result equals the one-bucket exchange up to the association of the sum
    ((g0 + g1) + (g2 + g3) across ranks first, against (g0 + g2) + (g1 + g3) locally first)."""
    P, world = 3000, 2
    out = str(tmp_path / "pipe")
    # n_kf = 3, 5: k % world sharding leaves rank 1 a keyframe short -- it must still issue rank 0's number of collectives
    # (a zeroed slot), or the ranks' collectives stop matching
    mp.spawn(_pipeline_worker, args=(world, _free_port(), P, n_kf, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["order"] == list(range((n_kf + 1) // 2)) and r1["order"] == list(range(n_kf // 2))
    for r in (r0, r1):
        assert torch.equal(r["overlap"], r["serial"])
    assert torch.equal(r0["overlap"], r1["overlap"]) and torch.equal(r0["bucket"], r1["bucket"])
    ref = torch.zeros(P * 14)
    for k in range(n_kf):
        ref += torch.cat([g.t().reshape(-1) for g in _keyframe_grads(P, k)[0]] + [torch.zeros(P * 2)])
    for name in ("overlap", "bucket"):
        assert torch.allclose(r0[name], ref, rtol=1e-5, atol=1e-5), name
    assert not torch.equal(r0["overlap"], r0["bucket"]) or True      # (association differs; equality is not required)
