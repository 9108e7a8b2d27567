"""Mapping-window sharding over 2 ranks on CPU (gloo): the all-reduced Gaussian gradients equal the
single-process sum over all keyframes, using the oracle as the per-keyframe renderer stand-in."""
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
