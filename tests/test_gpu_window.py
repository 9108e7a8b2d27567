"""The keyframe-sharded mapping window with REAL renders (BASELINE config C4), rehearsed with two
ranks that share cuda:0: every rank renders its keyframes with the HIP rasteriser, the Gaussian gradients / statistics /
visibility bits / poses travel over gloo (staged through host memory -- RCCL refuses two ranks on one device), and the
result is compared with the single-process window (sum over all keyframes, /root/reference/utils/slam_mapper.py:273-324,394).
"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_KF, ITERS = 4, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_window(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from monogs_amd import camera as cam
    from monogs_amd.gaussian_map import GaussianMap
    from monogs_amd.mapping import WindowMapper
    from monogs_amd.slam_harness import make_sequence
    from monogs_amd.window import replicas_in_sync

    dev = "cuda:0"
    torch.cuda.set_device(0)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, intr = make_sequence(N_KF, "fr3_office", n_gaussians=20000, device=dev)
    bg = torch.zeros(3, device=dev)
    gmap = GaussianMap(dev)
    for i, vp in enumerate(frames):          # keyframe poses start slightly off the truth: pose gradients are non-zero
        d = cam.se3_exp(torch.tensor([0.002 * i, -0.001 * i, 0.0, 0.0, 0.0005 * i, 0.0], device=dev))
        Tm = torch.eye(4, device=dev)
        Tm[:3, :3], Tm[:3, 3] = vp.R_gt, vp.T_gt
        Tn = d @ Tm
        vp.update_RT(Tn[:3, :3], Tn[:3, 3])
    gmap.extend_from_frame(frames[0], intr, downsample=8, init=True, point_size=1.0)
    before = [p.detach().clone() for p in gmap.params()]
    mapper = WindowMapper(gmap, intr, bg, window_size=N_KF)
    mapper.keep_reduced_grads = True
    mapper.optimize_map(frames, iters=1)
    grads = [g.cpu() for g in mapper.last_grads]
    vis = {k: v.cpu() for k, v in mapper.occ_aware_visibility.items()}
    stats = (gmap.xyz_gradient_accum.cpu().clone(), gmap.denom.cpu().clone(), gmap.max_radii_2d.cpu().clone())
    mapper.optimize_map(frames, iters=ITERS - 1)
    in_sync = replicas_in_sync(gmap.params())
    mapper.sync_poses(frames)
    torch.save(dict(grads=grads, vis=vis, stats=stats, params=[p.detach().cpu() for p in gmap.params()],
                    before=[b.cpu() for b in before], in_sync=in_sync,
                    poses=[(v.R.cpu(), v.T.cpu(), v.exposure_a.data.cpu(), v.exposure_b.data.cpu()) for v in frames],
                    moments=[m.cpu() for m in gmap.optimizer.exp_avg]), f"{out}.{world}.{rank}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_densify(rank, world, port, out):
    """Map surgery inside the sharded window: densify_and_prune on iterations 2 and 4 (clone + split with the seeded
    generator + prune), opacity reset of the non-visible on iteration 5, covisibility prune at the end."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from monogs_amd.gaussian_map import GaussianMap
    from monogs_amd.mapping import WindowMapper
    from monogs_amd.slam_harness import make_sequence
    from monogs_amd.window import replicas_in_sync

    dev = "cuda:0"
    torch.cuda.set_device(0)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, intr = make_sequence(N_KF, "fr3_office", n_gaussians=20000, device=dev)
    for vp in frames:
        vp.update_RT(vp.R_gt.clone(), vp.T_gt.clone())
    gmap = GaussianMap(dev)
    gmap.extend_from_frame(frames[0], intr, downsample=16, init=True, point_size=1.0)
    n0 = len(gmap)
    mapper = WindowMapper(gmap, intr, torch.zeros(3, device=dev), window_size=N_KF, seed=7)
    mapper.gaussian_update_every, mapper.gaussian_update_offset = 2, 0      # densify on iterations 2, 4
    mapper.gaussian_reset = 5                                                # opacity reset on iteration 5
    mapper.densify_grad_threshold = 1e-7                                     # so that clone AND split really fire
    mapper.gaussian_th = 0.05                                                # (0.7 would prune the whole young map: opacities start at 0.5)
    sizes = []
    for _ in range(5):
        mapper.optimize_map(frames, iters=1)
        sizes.append(len(gmap))
    had_state = any(int(po.t_dev.item()) > 0 for po in mapper._pose_opt.values())
    mapper.new_keyframe_optimizers(frames)                                   # a keyframe joins: fresh pose-optimiser state
    fresh = all(int(po.t_dev.item()) == 0 and float(po.m.abs().sum()) == 0 for po in mapper._pose_opt.values())
    mapper.optimize_map(frames, prune=True, iters=1)                         # full window: covisibility prune
    sizes.append(len(gmap))
    ok = replicas_in_sync(gmap.params() + [gmap.xyz_gradient_accum, gmap.denom, gmap.max_radii_2d,
                                           gmap.kf_idx.float(), gmap.nr_obs.float()] + gmap.optimizer.exp_avg)
    torch.save(dict(n0=n0, sizes=sizes, in_sync=ok, steps=gmap.optimizer.t_dev.cpu(), pose_reset=had_state and fresh,
                    vis_len=[int(v.shape[0]) for v in mapper.occ_aware_visibility.values()],
                    finite=all(bool(torch.isfinite(p).all()) for p in gmap.params())), f"{out}.d{world}.{rank}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_map_surgery_stays_in_sync(native_lib, tmp_path):
    """Densification (clone, seeded split, prune), opacity reset and covisibility pruning change the map size inside the
    sharded window; the replicas must go through them identically with no parameter broadcast."""
    out = str(tmp_path / "w")
    mp.spawn(_run_densify, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(f"{out}.d2.0"), torch.load(f"{out}.d2.1")
    assert r0["in_sync"] and r1["in_sync"] and r0["finite"]
    assert r0["sizes"] == r1["sizes"]
    assert r0["sizes"][1] != r0["n0"] and r0["sizes"][3] != r0["sizes"][2], r0["sizes"]      # both densify steps changed the map
    assert all(n == r0["sizes"][-1] for n in r0["vis_len"]) and len(r0["vis_len"]) == N_KF    # visibility re-indexed by the prune
    assert torch.equal(r0["steps"], r1["steps"])
    assert r0["pose_reset"] and r1["pose_reset"]


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def test_two_rank_window_matches_single_process(native_lib, tmp_path):
    out = str(tmp_path / "w")
    mp.spawn(_run_window, args=(2, _free_port(), out), nprocs=2, join=True)
    mp.spawn(_run_window, args=(1, 0, out), nprocs=1, join=True)
    r0, r1, one = (torch.load(f"{out}.{w}.{r}") for w, r in ((2, 0), (2, 1), (1, 0)))
    assert r0["in_sync"] and r1["in_sync"]
    # (1) replicas: bit-identical parameters, Adam moments and keyframe poses on both ranks after 3 iterations
    for a, b in zip(r0["params"] + r0["moments"], r1["params"] + r1["moments"]):
        assert torch.equal(a, b)
    for pa, pb in zip(r0["poses"], r1["poses"]):
        for a, b in zip(pa, pb):
            assert torch.equal(a, b)
    # (2) the all-reduced gradients equal the single-process sum over all keyframes (float atomics reorder sums: 1e-5)
    #     (isotropic scales: the rotation gradient is zero up to rounding noise, hence the absolute term)
    for a, b in zip(r0["grads"], one["grads"]):
        assert a.shape == b.shape
        assert (a.double() - b.double()).norm() <= 1e-5 * b.double().norm() + 1e-7, _rel(a, b)
    assert all(b.abs().max() > 0 for b in one["grads"][:4])
    # (3) statistics and visibility of iteration 1 (identical parameters on both sides): sums of per-keyframe norms,
    #     visible counts, MAX of radii, P visibility bits per keyframe
    assert _rel(r0["stats"][0], one["stats"][0]) < 1e-5
    assert torch.equal(r0["stats"][1], one["stats"][1]) and torch.equal(r0["stats"][2], one["stats"][2])
    assert sorted(r0["vis"]) == sorted(one["vis"]) == list(range(N_KF))
    for k in range(N_KF):
        assert torch.equal(r0["vis"][k], one["vis"][k]) and torch.equal(r0["vis"][k], r1["vis"][k])
        assert r0["vis"][k].any()
    # (4) after 3 Adam steps the update of every tensor agrees with the single-process update
    #     (not the rotations: Adam normalises their pure-noise gradient into full-size steps of random sign)
    for a, b, s in list(zip(r0["params"], one["params"], one["before"]))[:4]:
        assert _rel(a - s, b - s) < 2e-2, _rel(a - s, b - s)
    # (5) poses moved (keyframe 0 is the gauge and must not) and agree with the single-process window
    assert torch.equal(r0["poses"][0][0], one["poses"][0][0])
    for k in range(1, N_KF):
        assert _rel(r0["poses"][k][1], one["poses"][k][1]) < 1e-4
