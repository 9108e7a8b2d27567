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


# ---- round 3: the captured iteration, C4 at its own shape, the pruning call's carried gradients, RCCL ------------------
def _c4_setup(dev, n_kf, intrinsics, n_gaussians, downsample):
    from monogs_amd import camera as cam
    from monogs_amd.gaussian_map import GaussianMap, REFERENCE_LR_SCHEDULE
    from monogs_amd.slam_harness import make_sequence
    frames, intr = make_sequence(n_kf, intrinsics, n_gaussians=n_gaussians, device=dev)
    for i, vp in enumerate(frames):          # keyframe poses slightly off the truth: the pose gradients are non-zero
        d = cam.se3_exp(torch.tensor([0.002 * i, -0.001 * i, 0.0, 0.0, 0.0005 * i, 0.0], device=dev))
        Tm = torch.eye(4, device=dev)
        Tm[:3, :3], Tm[:3, 3] = vp.R_gt, vp.T_gt
        Tn = d @ Tm
        vp.update_RT(Tn[:3, :3], Tn[:3, 3])
    gmap = GaussianMap(dev)
    gmap.lr_schedule = dict(REFERENCE_LR_SCHEDULE, lr_init=gmap.lrs[0], lr_final=gmap.lrs[0] * 1e-3, max_steps=300)
    gmap.extend_from_frame(frames[0], intr, downsample=downsample, init=True, point_size=1.0)
    return frames, intr, gmap


def _run_c4(rank, world, port, out, use_graph, n_kf, intrinsics, n_gaussians, downsample, iters, tag, exchange="bucket"):
    """One mapping call over the whole window (BASELINE config C4 when 8 keyframes at 1200x680), sharded over `world`."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from monogs_amd.mapping import WindowMapper
    from monogs_amd.window import replicas_in_sync
    dev = "cuda:0"
    torch.cuda.set_device(0)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, intr, gmap = _c4_setup(dev, n_kf, intrinsics, n_gaussians, downsample)
    before = [p.detach().clone() for p in gmap.params()]
    mapper = WindowMapper(gmap, intr, torch.zeros(3, device=dev), window_size=n_kf, use_graph=use_graph)
    mapper.map_surgery = False
    mapper.exchange = exchange
    mapper.optimize_map(frames, iters=iters)
    in_sync = replicas_in_sync(gmap.params() + [gmap.xyz_gradient_accum, gmap.denom, gmap.max_radii_2d] + gmap.optimizer.exp_avg)
    mapper.sync_poses(frames)
    torch.save(dict(params=[p.detach().cpu() for p in gmap.params()], before=[b.cpu() for b in before], in_sync=in_sync,
                    stats=(gmap.xyz_gradient_accum.cpu(), gmap.denom.cpu(), gmap.max_radii_2d.cpu()),
                    vis={k: v.cpu() for k, v in mapper.occ_aware_visibility.items()}, nr_iters=mapper.nr_iters,
                    lrs=list(gmap.optimizer.lrs), mstats=dict(mapper.stats), P=len(gmap),
                    steps=gmap.optimizer.t_dev.cpu(), loss=float(mapper.last_loss) if mapper.last_loss is not None else None,
                    poses=[(v.R.cpu(), v.T.cpu(), v.exposure_a.data.cpu(), v.exposure_b.data.cpu()) for v in frames]),
               f"{out}.{tag}.{world}.{rank}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _check_windows_agree(a, b, n_kf, iters, tol_update=3e-2):
    """Two runs of the same window from the same start (different summation orders of float atomics / collectives)."""
    assert a["nr_iters"] == b["nr_iters"] == iters and a["P"] == b["P"]
    assert torch.equal(a["steps"], b["steps"]) and int(a["steps"][0]) == iters
    assert abs(a["lrs"][0] - b["lrs"][0]) <= 1e-7 * b["lrs"][0] and a["lrs"][0] < 0.99 * a["lrs"][1] * 0 + b["lrs"][0] * 1.0000001
    for x, y, s in list(zip(a["params"], b["params"], b["before"]))[:4]:       # (not the rotations: pure-noise gradients)
        assert (y - s).abs().max() > 0
        assert _rel(x - s, y - s) < tol_update, _rel(x - s, y - s)
    assert _rel(a["stats"][0], b["stats"][0]) < 1e-3                            # sums of per-keyframe gradient norms
    assert (a["stats"][1] != b["stats"][1]).float().mean() < 1e-3               # visible counts (a radius may flip at 0 / 1)
    assert (a["stats"][2] != b["stats"][2]).float().mean() < 1e-3
    assert sorted(a["vis"]) == sorted(b["vis"]) == list(range(n_kf))
    for k in range(n_kf):
        assert (a["vis"][k] != b["vis"][k]).float().mean() < 1e-3 and a["vis"][k].any()
    assert torch.equal(a["poses"][0][0], b["poses"][0][0])                      # keyframe 0 is the gauge
    for k in range(1, n_kf):
        assert _rel(a["poses"][k][1], b["poses"][k][1]) < 1e-3


def test_captured_window_iteration_equals_eager(native_lib, tmp_path):
    """The hipGraph-replayed mapping iteration (one eager iteration, one capture, replays) against the same iterations run
    eagerly: same code (`WindowMapper._front` / `_back`), so the only difference left is summation order; the iteration
    count, the Adam step counts and the device-stepped xyz learning rate agree exactly."""
    out, iters = str(tmp_path / "g"), 12
    args = (4, "fr3_office", 20000, 8, iters)
    mp.spawn(_run_c4, args=(1, 0, out, False) + args + ("e",), nprocs=1, join=True)
    mp.spawn(_run_c4, args=(1, 0, out, True) + args + ("g",), nprocs=1, join=True)
    e, g = torch.load(f"{out}.e.1.0"), torch.load(f"{out}.g.1.0")
    assert g["mstats"]["captures"] == 1 and g["mstats"]["replays"] == iters - 1 and g["mstats"]["eager_iters"] == 1
    assert e["mstats"]["replays"] == 0 and e["mstats"]["eager_iters"] == iters
    _check_windows_agree(g, e, 4, iters)
    from monogs_amd.gaussian_optim import expon_lr
    from monogs_amd.gaussian_map import DEFAULT_LRS, REFERENCE_LR_SCHEDULE
    want = expon_lr(iters, **dict(REFERENCE_LR_SCHEDULE, lr_init=DEFAULT_LRS[0], lr_final=DEFAULT_LRS[0] * 1e-3, max_steps=300))
    assert abs(g["lrs"][0] - want) <= 1e-6 * want and abs(e["lrs"][0] - want) <= 1e-6 * want    # update_learning_rate(nr_iters)


def test_c4_window_two_ranks_match_one_rank_captured(native_lib, tmp_path):
    """BASELINE config C4 at its own shape: an 8-keyframe window at Replica resolution (1200x680,
    /root/reference/configs/rgbd/replica/base_config.yaml:27-28,39-48), ~100 k Gaussians, keyframes sharded over 2 ranks
    (gloo staged through host memory: RCCL refuses two ranks on one device), both runs on captured iterations."""
    out, iters, n_kf = str(tmp_path / "c4"), 10, 8
    args = (n_kf, "replica", 150000, 8, iters)
    mp.spawn(_run_c4, args=(2, _free_port(), out, True) + args + ("c",), nprocs=2, join=True)
    mp.spawn(_run_c4, args=(1, 0, out, True) + args + ("c",), nprocs=1, join=True)
    r0, r1, one = (torch.load(f"{out}.c.{w}.{r}") for w, r in ((2, 0), (2, 1), (1, 0)))
    assert r0["P"] > 80000
    assert r0["in_sync"] and r1["in_sync"]
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)                                   # replicas bit-identical, no parameter broadcast
    for pa, pb in zip(r0["poses"], r1["poses"]):
        for a, b in zip(pa, pb):
            assert torch.equal(a, b)                               # after sync_poses
    assert r0["mstats"]["captures"] == 1 and r0["mstats"]["replays"] == iters - 1
    _check_windows_agree(r0, one, n_kf, iters)


@pytest.mark.parametrize("use_graph,n_kf", [(False, 4), (True, 4), (False, 3), (True, 3)])
def test_pipelined_exchange_window_matches_single_process(native_lib, tmp_path, use_graph, n_kf):
    """`exchange = "per_keyframe"`: with two keyframes per rank every owned keyframe's gradients are all-reduced on their own,
    issued behind its backward while the next keyframe renders (eager: one (render, loss, backward) per keyframe; captured: one
    graph per keyframe, the collectives between the replays), then added in keyframe order.  Against the single-process
    window the result differs by summation order only; the two replicas hold the same bits."""
    # (n_kf = 3: rank 1 owns ONE keyframe, rank 0 two -- both must issue two slot collectives of the same size)
    out, iters = str(tmp_path / "pk"), 10
    args = (n_kf, "fr3_office", 20000, 8, iters)
    mp.spawn(_run_c4, args=(2, _free_port(), out, use_graph) + args + ("p", "per_keyframe"), nprocs=2, join=True)
    mp.spawn(_run_c4, args=(1, 0, out, use_graph) + args + ("p",), nprocs=1, join=True)
    r0, r1, one = (torch.load(f"{out}.p.{w}.{r}") for w, r in ((2, 0), (2, 1), (1, 0)))
    assert r0["in_sync"] and r1["in_sync"]
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    if use_graph:
        assert r0["mstats"]["captures"] == 1 and r0["mstats"]["replays"] == iters - 1
    _check_windows_agree(r0, one, n_kf, iters)


def _run_prune_carry(rank, world, port, out):
    """A pruning call on a window that is NOT full keeps its gradients (no optimiser step, nothing zeroes them); the next
    iteration's backward adds to them (/root/reference/utils/slam_mapper.py:394-451,482-483).  Sharded: they must be summed
    over the ranks ONCE."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from monogs_amd.mapping import WindowMapper
    dev = "cuda:0"
    torch.cuda.set_device(0)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, intr, gmap = _c4_setup(dev, 3, "fr3_office", 20000, 8)
    mapper = WindowMapper(gmap, intr, torch.zeros(3, device=dev), window_size=8)       # window of 3 < 8: not full
    mapper.map_surgery = False
    mapper.keep_reduced_grads = True
    mapper.optimize_map(frames, iters=1)
    plain = [g.cpu() for g in mapper.last_grads]
    n_before = len(gmap)
    mapper.optimize_map(frames, prune=True, iters=1)
    vis_after_prune = {k: v.cpu() for k, v in mapper.occ_aware_visibility.items()}
    steps_after_prune = gmap.optimizer.t_dev.cpu().clone()
    mapper.optimize_map(frames, iters=1)
    carried = [g.cpu() for g in mapper.last_grads]
    mapper.optimize_map(frames, iters=1)
    after = [g.cpu() for g in mapper.last_grads]
    torch.save(dict(plain=plain, carried=carried, after=after, n=(n_before, len(gmap)), vis=vis_after_prune,
                    steps=(steps_after_prune, gmap.optimizer.t_dev.cpu()), nr_iters=mapper.nr_iters), f"{out}.{world}.{rank}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_pruning_call_gradients_are_carried_once(native_lib, tmp_path):
    out = str(tmp_path / "p")
    mp.spawn(_run_prune_carry, args=(2, _free_port(), out), nprocs=2, join=True)
    mp.spawn(_run_prune_carry, args=(1, 0, out), nprocs=1, join=True)
    r0, r1, one = (torch.load(f"{out}.{w}.{r}") for w, r in ((2, 0), (2, 1), (1, 0)))
    assert one["n"][0] == one["n"][1] and one["nr_iters"] == 4                 # window not full: nothing pruned
    assert int(one["steps"][0][0]) == 1 and int(one["steps"][1][0]) == 3      # the pruning call takes no optimiser step
    assert len(one["vis"]) == 3 and all(v.any() for v in one["vis"].values())
    for c, a, pl in zip(one["carried"][:4], one["after"][:4], one["plain"][:4]):
        # the iteration after the pruning call steps on (pruning call's gradient + its own) ~ twice a plain gradient;
        # the one after that is plain again
        assert 1.6 < (c.norm() / a.norm()).item() < 2.4, (c.norm() / a.norm()).item()
        assert 0.7 < (a.norm() / pl.norm()).item() < 1.4
    for k in ("plain", "carried", "after"):
        for a, b, c in zip(r0[k], r1[k], one[k]):
            assert torch.equal(a, b)
            assert (a.double() - c.double()).norm() <= 2e-4 * c.double().norm() + 1e-7, (k, _rel(a, c))


@pytest.mark.gpu
def test_full_window_pruning_call_keeps_pose_gradients(native_lib):
    """slam_mapper.py:408-451 returns right after prune_points: the Gaussian gradients go with the replaced tensors, the pose /
    exposure gradients of the pruning iteration STAY in .grad (nothing zeroes them) and join the first step of the next call."""
    from monogs_amd.mapping import WindowMapper
    dev = "cuda:0"
    frames, intr, gmap = _c4_setup(dev, 3, "fr3_office", 20000, 8)
    mapper = WindowMapper(gmap, intr, torch.zeros(3, device=dev), window_size=3)        # window of 3 == 3: full
    mapper.map_surgery = False
    mapper.prune_coviz = 1
    mapper.optimize_map(frames, iters=2)
    for vp in frames:
        assert vp.cam_rot_delta.grad is None and vp.exposure_a.grad is None             # zeroed after a normal iteration
    mapper.optimize_map(frames, prune=True, iters=1)
    moved = [vp for vp in frames if vp.frame_idx != 0]
    for vp in moved:
        for q in (vp.cam_rot_delta, vp.cam_trans_delta, vp.exposure_a, vp.exposure_b):
            assert q.grad is not None and bool(torch.isfinite(q.grad).all())
        assert float(vp.cam_rot_delta.grad.abs().sum()) > 0
    assert all(q.grad is None for q in gmap.params())
    kept = [vp.cam_trans_delta.grad.clone() for vp in moved]
    mapper.keep_reduced_grads = True
    mapper.optimize_map(frames, iters=1)               # takes them in (eager first iteration), steps, zeroes
    for vp in moved:
        assert vp.cam_trans_delta.grad is None
    assert all(bool(torch.isfinite(k).all()) for k in kept)


def _run_rccl_one_rank(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from monogs_amd.mapping import WindowMapper
    from monogs_amd.window import replicas_in_sync
    dev = "cuda:0"
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
    frames, intr, gmap = _c4_setup(dev, 3, "fr3_office", 20000, 8)
    res = {}
    for graph in (False, True):
        mapper = WindowMapper(gmap, intr, torch.zeros(3, device=dev), window_size=3, use_graph=graph, force_collectives=True)
        mapper.map_surgery = False
        mapper.keep_reduced_grads = True
        mapper.prune_coviz = 1                   # (three keyframes: the reference's "seen by at most 3" would drop the whole map)
        assert mapper.sharded and mapper.world == 1
        mapper.optimize_map(frames, iters=10 if graph else 2)
        mapper.optimize_map(frames, prune=True, iters=1)                   # full window: all-gather of the bits only, then prune
        mapper.sync_poses(frames)                                           # all-gather of 14 floats per keyframe
        res[graph] = dict(sync=replicas_in_sync(gmap.params(), force=True),                       # int64 MIN / MAX
                          sync_bad=replicas_in_sync([torch.full((3,), float("nan"), device=dev)], force=True),
                          vis=[bool(v.any()) for v in mapper.occ_aware_visibility.values()], P=len(gmap),
                          stats=dict(mapper.stats), finite=all(bool(torch.isfinite(p).all()) for p in gmap.params()),
                          accum=float(gmap.xyz_gradient_accum.abs().sum()), maxr=float(gmap.max_radii_2d.max()))
    torch.save(dict(res=res, backend=dist.get_backend()), out)
    dist.barrier()
    dist.destroy_process_group()


def test_exchanges_run_on_rccl_with_one_rank(native_lib, tmp_path):
    """Every collective of the sharded window executed by the RCCL backend (`nccl`) in a one-rank group on cuda:0: the
    in-place all-reduce(SUM) of the flat gradient bucket, the int64 all-gather carrying MAX radii + visibility bits, the
    pose all-gather, the uint8 MAX of the opacity-reset union, the int64 MIN / MAX of `replicas_in_sync` -- eager and
    between the two captured halves of the iteration."""
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_run_rccl_one_rank, args=(1, _free_port(), out), nprocs=1, join=True)
    r = torch.load(out)
    assert r["backend"] == "nccl"
    for graph in (False, True):
        x = r["res"][graph]
        assert x["sync"] and x["finite"] and all(x["vis"]) and x["accum"] > 0 and x["maxr"] > 0
        assert x["sync_bad"]            # (one rank always agrees with itself: the reductions ran and returned)
    assert r["res"][True]["stats"]["captures"] >= 1 and r["res"][True]["stats"]["replays"] >= 8
