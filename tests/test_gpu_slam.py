"""BASELINE config 3 in miniature: the tracking and mapping loops that drive the rasteriser
(/root/reference/utils/slam_tracker.py:138-188, /root/reference/utils/slam_mapper.py:244-500) on a short synthetic
TUM-like RGB-D sequence (fr3_office intrinsics; the dataset is not available offline), eager and hipGraph-captured."""
import pytest
import torch

pytestmark = pytest.mark.gpu
CFG = dict(n_frames=5, intrinsics="fr3_office", tracking_itr_num=100, mapping_itr_num=30, window_size=8, kf_interval=2,
           init_itr_num=80, n_gaussians=30000)


@pytest.fixture(scope="module")
def runs(native_lib):
    from monogs_amd.slam_harness import run_slam
    eager = run_slam(graph_tracking=False, graph_mapping=False, **CFG)
    graph = run_slam(graph_tracking=True, graph_mapping=False, **CFG)
    return eager, graph


def test_tracking_converges_and_reduces_the_pose_error(runs):
    eager, graph = runs
    assert graph["ate_rmse_m"] < 2e-3, graph["ate_rmse_m"]          # the end-to-end run on captured tracking iterations
    c, gt = eager["camera_centers"], eager["camera_centers_gt"]
    for i in range(1, CFG["n_frames"]):
        before = (c[i - 1] - gt[i]).norm().item()       # tracking starts from the previous frame's estimate
        after = (c[i] - gt[i]).norm().item()
        assert after < 0.5 * before, (i, before, after)
    assert eager["ate_rmse_m"] < 2e-3, eager["ate_rmse_m"]
    # the early exit of slam_tracker.py:172-176 fires: no frame needs all 100 iterations
    assert all(1 < n < CFG["tracking_itr_num"] for _, n in eager["track_iters_per_frame"]), eager["track_iters_per_frame"]


def test_graph_tracking_equals_eager_tracking(native_lib):
    """A captured tracking iteration replayed until the device-side convergence flag rises = the eager loop with its
    per-iteration `if converged: break` (/root/reference/utils/slam_tracker.py:138-188).  Like with like: same map, same
    frame, same start pose, and BOTH loops render the frozen map (no gradient for the Gaussians), so both run the six-sum
    pose-only blend backward and the only difference left is the order of its float atomics.  Poses agree to 1e-5.  The
    iteration COUNT is not compared: near the end the Adam step hovers around the 1e-4 exit threshold for several
    iterations and the last bits of the gradient decide which of them dips below it first (69 vs 65 was observed) -- instead
    the convergence logic is checked exactly: the flag is up, and the device-side Adam step count equals the iterations the
    graph loop reports (a surplus replay after convergence is a no-op: the flag is sticky)."""
    from monogs_amd import fused_losses
    from monogs_amd.gaussian_map import GaussianMap
    from monogs_amd.mapping import WindowMapper
    from monogs_amd.pose_optim import PoseAdam
    from monogs_amd.renderer import render
    from monogs_amd.slam_harness import TrackingGraph, Viewpoint, make_sequence
    dev = "cuda:0"
    frames, intr = make_sequence(3, "fr3_office", n_gaussians=30000, device=dev)
    bg = torch.zeros(3, device=dev)
    gmap = GaussianMap(dev)
    frames[0].update_RT(frames[0].R_gt.clone(), frames[0].T_gt.clone())
    gmap.extend_from_frame(frames[0], intr, downsample=8, init=True, point_size=1.0)
    mapper = WindowMapper(gmap, intr, bg, window_size=8)
    mapper.map_surgery = False                      # (the mapping thresholds would prune this 80-iteration-old map)
    mapper.optimize_map([frames[0]], iters=80, init=True)                                            # one map for both loops
    with torch.no_grad():
        frozen = (gmap.get_xyz.detach(), gmap.get_rotation.detach(), gmap.get_scaling.detach(), gmap.get_opacity.detach(),
                  gmap.get_features.detach())

    def start(i):                     # frame i, starting from the previous frame's true pose (as the tracker does)
        f = frames[i]
        vp = Viewpoint(f.frame_idx, f.rgb, f.depth, dev, gt_R=f.R_gt, gt_T=f.T_gt)
        vp.update_RT(frames[i - 1].R_gt.clone(), frames[i - 1].T_gt.clone())
        return vp

    for i in (1, 2):
        # ---- eager: render -> get_loss_tracking -> backward -> Adam step + update_pose, leave when converged
        ve = start(i)
        opt = PoseAdam(ve, 0.003, 0.001, 0.01)
        n_eager = 100
        for it in range(100):
            pkg = render(ve, intr, *frozen, bg)
            opt.zero_grad()
            fused_losses.get_loss_tracking(pkg["render"], pkg["depth"], pkg["opacity"], ve).backward()
            with torch.no_grad():
                if opt.step_and_retract():
                    n_eager = it + 1
                    break
        # ---- hipGraph: the same iteration captured once, replayed until the sticky device flag rises
        vg = start(i)
        tg = TrackingGraph(vg, intr, gmap, bg)
        n_graph = tg.track(vg, 100)
        flag_up, steps = float(tg.opt.out[0]) > 0.5, int(tg.opt.t_dev.item())
        tg.close()
        assert 1 < n_eager < 100 and 1 < n_graph < 100, (i, n_eager, n_graph)
        assert flag_up and steps == n_graph, (i, flag_up, steps, n_graph)
        assert (ve.R - vg.R).abs().max() < 1e-5 and (ve.T - vg.T).abs().max() < 1e-5, (i, (ve.R - vg.R).abs().max())
        err0 = (-(frames[i - 1].R_gt.t() @ frames[i - 1].T_gt) + (frames[i].R_gt.t() @ frames[i].T_gt)).norm()
        err1 = (-(vg.R.t() @ vg.T) + (frames[i].R_gt.t() @ frames[i].T_gt)).norm()
        assert err1 < 0.5 * err0, (i, float(err0), float(err1))


def test_mapping_reduces_its_loss(runs):
    eager, _ = runs
    assert len(eager["map_loss"]) >= 2
    first, last = eager["map_loss"][0]                   # map initialisation: the loss must fall a lot
    assert last < 0.6 * first, eager["map_loss"]
    for first, last in eager["map_loss"][1:]:            # window optimisations start near the optimum: must not rise
        assert last <= first * 1.02, eager["map_loss"]


def test_graph_mapping_runs_clean(native_lib):
    """Whole mapping iterations captured in a hipGraph (capacity mode): same trajectory quality, no overflow flag
    (run_slam raises on a raised flag)."""
    from monogs_amd import rasterizer as _r
    from monogs_amd.slam_harness import run_slam
    r = run_slam(graph_tracking=True, graph_mapping=True, **CFG)
    assert r["ate_rmse_m"] < 2e-3
    assert r["mapping_steady_iters_per_s"] and r["mapping_steady_iters_per_s"] > 0
    assert not _r.check_overflow()


def test_two_process_run_hands_the_map_over_through_the_arena(native_lib):
    """The reference's process topology (/root/reference/slam.py:102-179): tracker here, mapper in a spawned process, the
    map crossing the boundary on every keyframe -- through `MapArena` (device buffers shared once over HIP IPC) instead of
    `clone_obj` + `mp.Queue` (/root/reference/utils/slam_mapper.py:550-564).  Same trajectory quality as the one-process run;
    every snapshot the tracker used was consistent."""
    from monogs_amd.slam_harness import run_slam_two_process
    r = run_slam_two_process(n_frames=7, intrinsics="fr3_office", tracking_itr_num=100, mapping_itr_num=30, window_size=8,
                             kf_interval=2, init_itr_num=80, n_gaussians=30000)
    assert r["exitcode"] == 0
    assert r["keyframes"] == 4 and r["sequences"] == [1, 2, 3, 4] and r["window_sizes"] == [1, 2, 3, 4]
    assert r["gaussians"][0] > 1000 and all(b >= a for a, b in zip(r["gaussians"], r["gaussians"][1:]))   # the map only grows
    assert r["tracked"] == 6 and r["ate_rmse_m"] < 3e-3, r["ate_rmse_m"]
    assert r["handoff_ms"]["publish"] < 50 and r["handoff_ms"]["acquire"] < 5, r["handoff_ms"]


def test_tracking_and_mapping_with_the_reference_map_surgery(native_lib):
    """BASELINE config 3 with EVERYTHING optimize_map does (/root/reference/utils/slam_mapper.py:408-451,462-480): densify_and_prune
    every 150 iterations at the 0.7 opacity threshold, covisibility pruning after each keyframe once the window is full,
    the fork's new-Gaussian recipe (1/32 and 1/64 of the pixels, scale^2 = dist2 x min(0.05, 0.01 x median depth)) and its
    learning rates -- on a sequence of OPAQUE surfaces (a ray-cast room), hipGraph-replayed with a re-capture after every
    map-size change.  The map must neither collapse nor explode, both halves of the surgery must fire, and tracking must
    keep converging against the maps the surgery leaves behind.

    How good "keeps converging" has to be is DERIVED, not read off a run (round 4 widened fitted bars after a failure):
      * against the same 13 frames run EAGERLY (no hipGraph, exact instance counts: the loop an unmodified caller runs): the
        replayed run's worst frame and its ATE may be at most twice the eager run's (+ 1.5 mm, 15 % of the ~1 cm the camera
        moves per frame: two runs of one sequence differ by the order of their float atomics, and the surgery's thresholds
        turn that into slightly different maps; measured in round 5: 9.9 / 3.0 mm replayed against 11.0 / 3.3 mm eager);
      * against the same frames with the surgery switched OFF: pruning may cost accuracy, but not more than a factor of two
        (+ 1.5 mm) in ATE -- the reference prunes on every keyframe and still tracks (measured: 3.0 against 1.3 mm).
    The worst frame is the one right behind a covisibility prune: it drops every Gaussian that at most three of the window's
    keyframes touch with T (1 - alpha) > 0.5 -- about two thirds of the map at once, mostly clones sitting behind their originals,
    but also the only Gaussians of regions that just came into view -- so that frame is tracked against a map with holes, where the
    tracking loss's opacity weighting leaves fewer pixels to hold the pose."""
    from monogs_amd.slam_harness import run_slam
    cfg = dict(n_frames=13, intrinsics="fr3_office", tracking_itr_num=100, mapping_itr_num=150, window_size=5, kf_interval=2,
               init_itr_num=1050, scene="room", reference_densify=True, reference_lrs=True)
    r = run_slam(map_surgery=True, graph_tracking=True, graph_mapping=True, **cfg)
    s = r["surgery"]
    sizes = s["gaussians_after_keyframe"]
    print("map size after each keyframe:", sizes, {k: v for k, v in s.items() if k not in ("log", "gaussians_after_keyframe")})
    assert r["map_surgery"] and len(sizes) == 7 and r["window_sizes"][-1] == 5
    assert s["densify_and_prune_calls"] >= 14                      # 11 of initialize_map + one per 150 mapping iterations
    assert s["cloned"] + s["split_net"] > 5000 and s["calls_that_grew"] >= 10        # densification fires ...
    assert s["pruned"] > 1000 and s["calls_that_pruned"] >= 1                         # ... and so does the opacity prune
    assert s["covisibility_prunes"] >= 2                           # the window of five is full from the fifth keyframe on
    assert min(sizes) > 4000 and max(sizes) < 150000, sizes        # 9 600 initial Gaussians: no collapse, no explosion
    assert r["mapping_captures"] >= 6 and r["mapping_replays"] > 1500
    assert all(1 < n < 100 for _, n in r["track_iters_per_frame"]), r["track_iters_per_frame"]     # every frame converges before the cap
    first, last = r["map_loss"][0]
    assert last < 0.1 * first                                      # map initialisation from sparse dots to a covered image

    eager = run_slam(map_surgery=True, graph_tracking=False, graph_mapping=False, **cfg)
    # (the surgery-free reference maps EAGERLY: without the new plan that surgery forces every <= 150 iterations, the replays of one
    #  captured initialisation iteration outgrow the instance capacity recorded at capture time -- the sparse initial dots triple
    #  their footprint within a few hundred iterations -- and the harness, rightly, raises instead of dropping instances, also with
    #  WindowMapper.max_replays_per_capture = 256 refreshing the capacities; this run is an accuracy reference, its speed is irrelevant)
    off = run_slam(map_surgery=False, graph_tracking=True, graph_mapping=False, **cfg)
    worst = lambda x: max(x["position_error_m"])  # noqa: E731
    print(f"replayed + surgery: worst frame {worst(r) * 1e3:.2f} mm, ATE {r['ate_rmse_m'] * 1e3:.2f} mm | eager + surgery: "
          f"{worst(eager) * 1e3:.2f} / {eager['ate_rmse_m'] * 1e3:.2f} mm | no surgery: {worst(off) * 1e3:.2f} / "
          f"{off['ate_rmse_m'] * 1e3:.2f} mm")
    assert eager["surgery"]["covisibility_prunes"] >= 2
    slack = 1.5e-3
    assert worst(r) <= 2.0 * worst(eager) + slack and r["ate_rmse_m"] <= 2.0 * eager["ate_rmse_m"] + slack
    assert r["ate_rmse_m"] <= 2.0 * off["ate_rmse_m"] + slack
    assert worst(eager) < 0.05 and worst(off) < 0.05             # (sanity of the references themselves: centimetres, not decimetres)
