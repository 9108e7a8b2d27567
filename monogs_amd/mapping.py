"""``optimize_map`` -- the mapping-window optimisation that drives the rasteriser, optionally sharded by keyframe
over the GPUs of a node (BASELINE.json config C4: 8-keyframe window, one keyframe per GPU).

Mirror of ``Mapper.optimize_map`` (/root/reference/utils/slam_mapper.py:244-500): per iteration render every window
keyframe against the shared Gaussians (:273-324), sum ``get_loss_mapping``, ONE backward (:394), occlusion-aware
visibility per keyframe (:400-404), covisibility pruning on request (:408-448), ``max_radii_2d`` + densification
statistics per keyframe (:453-460), ``densify_and_prune`` every ``gaussian_update_every`` iterations (:462-473),
opacity reset of non-visible Gaussians (:476-479), Adam step on the Gaussians and on the keyframe poses +
``update_pose`` (:482-496).

Sharding (`monogs_amd.window`): rank r renders the window positions ``k % world == r``; the
Gaussians are replicated.  Per iteration the ranks exchange
  * ONE all-reduce(SUM) of the Gaussian gradients (12 floats per Gaussian) with the two densification statistics
    (sum over keyframes of the per-keyframe screen-space gradient norm, and of the visibility count) riding along,
  * one all-reduce(MAX) of ``max_radii_2d``,
  * an all-gather of P visibility bits per keyframe,
and afterwards every rank applies the same Adam step and the same (identically seeded) map surgery, so the replicas
stay bit-identical with no broadcast of parameters.  Pose / exposure parameters and their optimiser state live on
the owning rank only; ``sync_poses`` all-gathers them before the map is handed on (:553-556).
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import fused_losses, window as W
from .gaussian_map import GaussianMap
from .gaussian_optim import activate, add_densification_stats, fan_out
from .pose_optim import PoseAdam
from .renderer import render


def render_map(vp, intr, gmap: GaussianMap, bg):
    """``render()`` with the map's activations (normalize / exp / sigmoid, forward and backward) in one launch each."""
    if gmap.fused_adam and gmap._rotation.requires_grad:
        rot, scales3, opac = activate(gmap._rotation, gmap._scaling, gmap._opacity)
        return render(vp, intr, gmap.get_xyz, rot, scales3, opac, gmap.get_features, bg)
    return render(vp, intr, gmap.get_xyz, gmap.get_rotation, gmap.get_scaling, gmap.get_opacity, gmap.get_features, bg)


class WindowMapper:
    # the values hard-coded in the fork's mapper (/root/reference/utils/slam_mapper.py:65-89)
    gaussian_update_every = 150
    gaussian_update_offset = 50
    gaussian_th = 0.7
    gaussian_extent = 1.0
    gaussian_reset = 2001
    size_threshold = 20
    densify_grad_threshold = 0.0002      # /root/reference/configs/mono/tum/base_config.yaml:66
    prune_coviz = 3

    def __init__(self, gmap: GaussianMap, intr, bg, group=None, window_size: int = 8, seed: int = 0,
                 lr_rot: float = 0.003 * 0.5, lr_trans: float = 0.001 * 0.5, lr_exposure: float = 0.01,
                 loss_fn=None):
        self.gmap, self.intr, self.bg, self.group = gmap, intr, bg, group
        self.window_size, self.seed = int(window_size), int(seed)
        self.world, self.rank = W._world(group), W._rank(group)
        self.lrs = (lr_rot, lr_trans, lr_exposure)
        self.loss_fn = loss_fn or fused_losses.get_loss_mapping
        self.nr_iters = 0
        self.first_time_pruned = False
        self.occ_aware_visibility: Dict[int, torch.Tensor] = {}     # kf id -> bool[P]
        self._pose_opt: Dict[int, PoseAdam] = {}                    # id(viewpoint) -> optimiser state (owning rank)
        self.last_loss = None
        self.exposed_comm_s = 0.0       # wall time spent waiting in collectives (diagnostic, synchronises when on)
        self.time_comm = False
        self._bucket = None
        self.map_surgery = True          # False: no densify_and_prune / opacity reset (fixed-size workloads: benchmarks, tests)
        self.keep_reduced_grads = False  # tests: clones of the (all-reduced) Gaussian gradients of the last iteration
        self.parallel_keyframes = False  # True: render / back-propagate the owned keyframes on a stream each.  Pays inside a
                                         # captured iteration (slam_harness: 863 -> 1254 it/s); this eager loop is bound by
                                         # the host issuing its ~300 launches (C4 on one GPU: 5.04 vs 5.14 ms), so it is off
        self._streams: List = []
        self.last_grads = None

    # ---- per-keyframe optimiser state lives on the owning rank ------------------------------------------------------
    def _pose_optimizer(self, vp) -> PoseAdam:
        po = self._pose_opt.get(id(vp))
        if po is None:
            po = self._pose_opt[id(vp)] = PoseAdam(vp, *self.lrs)
        return po

    def _kf_streams(self, n: int):
        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream(device=self.gmap.device))
        return self._streams[:n]

    def new_keyframe_optimizers(self, viewpoints: Sequence = ()):
        """What the mapper does whenever a keyframe joins the window: ``self.keyframe_optimizers = torch.optim.Adam(...)``
        over the window's pose / exposure parameters (/root/reference/utils/slam_mapper.py:669-719) -- a FRESH optimiser,
        i.e. zero moments and step counts for every keyframe of the window.  Call it before the ``optimize_map`` calls
        of a new keyframe; between them (``prune=False`` then ``prune=True``) the state carries over, as there."""
        self._pose_opt = {k: v for k, v in self._pose_opt.items() if any(k == id(vp) for vp in viewpoints)}
        for po in self._pose_opt.values():
            po.reset()

    def owned(self, n_keyframes: int) -> List[int]:
        return W.shard_keyframes(n_keyframes, self.rank, self.world)

    # ---- one call = Mapper.optimize_map(cur_kf_list, prune, iters) ------------------------------------------------------
    def optimize_map(self, viewpoints: Sequence, kf_ids: Optional[Sequence[int]] = None, prune: bool = False,
                     iters: int = 1, init: bool = False) -> bool:
        """``viewpoints``: the window's keyframes (every rank holds all of them; only the owned ones are rendered).
        ``kf_ids``: their keyframe ids (default: ``vp.frame_idx``).  Returns ``gaussian_split`` as the reference."""
        n = len(viewpoints)
        if n == 0:
            return False
        kf_ids = [int(v.frame_idx) for v in viewpoints] if kf_ids is None else [int(k) for k in kf_ids]
        gmap = self.gmap
        mine = self.owned(n)
        gaussian_split = False
        for _ in range(iters):
            self.nr_iters += 1
            P = len(gmap)
            pkgs = {}
            loss = None
            fans = None
            if gmap.fused_adam and gmap._rotation.requires_grad and len(mine) > 1:
                # activations once per iteration, and each render through its own aliases of the five map tensors: their
                # gradients then meet in ONE node that adds them in one launch (gaussian_optim.fan_out)
                rot, scales3, opac = activate(gmap._rotation, gmap._scaling, gmap._opacity)
                fans = fan_out(len(mine), gmap.get_xyz, gmap.get_features, opac, scales3, rot)
            # the owned keyframes are independent until their losses are added: a stream each (the small latency-bound
            # kernels of one render overlap the blend kernels of another)
            streams = self._kf_streams(len(mine)) if (self.parallel_keyframes and len(mine) > 1
                                                      and torch.device(gmap.device).type == "cuda") else None
            main = torch.cuda.current_stream() if streams else None
            terms = []
            for j, k in enumerate(mine):
                if streams:
                    streams[j].wait_stream(main)
                with (torch.cuda.stream(streams[j]) if streams else contextlib.nullcontext()):
                    if fans is not None:
                        xyz_k, feat_k, opac_k, sc_k, rot_k = fans[j]
                        pkg = render(viewpoints[k], self.intr, xyz_k, rot_k, sc_k, opac_k, feat_k, self.bg)
                    else:
                        pkg = render_map(viewpoints[k], self.intr, gmap, self.bg)
                    if pkg is None:
                        raise ValueError("Render package is None")
                    terms.append(self.loss_fn(pkg["render"], pkg["depth"], viewpoints[k], init=init))
                pkgs[k] = pkg
            if streams:
                for st in streams:
                    main.wait_stream(st)
            for term in terms:
                loss = term if loss is None else loss + term
            if loss is not None:
                loss.backward()
            self.last_loss = loss

            with torch.no_grad():
                # ---- local statistics of the keyframes rendered here (per-keyframe norm BEFORE any summation)
                d_norm = torch.zeros(P, 1, device=gmap.device)
                d_vis = torch.zeros(P, 1, device=gmap.device)
                d_maxr = torch.zeros(P, device=gmap.device)
                for k in mine:
                    add_densification_stats(pkgs[k]["viewspace_points"].grad, pkgs[k]["radii"], d_norm, d_vis, d_maxr)
                # ---- exchanges
                if self.world > 1:
                    d_norm, d_vis = self._exchange(d_norm, d_vis, d_maxr)
                vis = W.all_gather_visibility({k: pkgs[k]["n_touched"] for k in mine}, n, P, self.group)
                self.occ_aware_visibility = {kf_ids[k]: vis[k] for k in range(n)}

                if prune:
                    # (as the reference: no optimiser step on a pruning call; when the window is not full yet the
                    #  gradients of this iteration stay in .grad and the next call's backward adds to them)
                    if n == self.window_size:
                        self._prune_covisibility(kf_ids)
                    return False

                torch.maximum(gmap.max_radii_2d, d_maxr, out=gmap.max_radii_2d)
                gmap.xyz_gradient_accum += d_norm
                gmap.denom += d_vis

                update_gaussian = self.map_surgery and \
                    self.nr_iters % self.gaussian_update_every == self.gaussian_update_offset
                if update_gaussian:
                    gmap.densify_and_prune(self.densify_grad_threshold, self.gaussian_th, self.gaussian_extent,
                                           self.size_threshold,
                                           generator=W.split_generator(gmap.device, self.seed, self.nr_iters))
                    gaussian_split = True
                if self.map_surgery and (self.nr_iters % self.gaussian_reset) == 0 and not update_gaussian:
                    # every keyframe's visibility_filter (radii > 0); only their union matters
                    gmap.reset_opacity_nonvisible([self._union_visible(pkgs, mine, P)])
                    gaussian_split = True

                if self.keep_reduced_grads:
                    self.last_grads = [None if p.grad is None else p.grad.clone() for p in gmap.params()]
                gmap.optimizer.step()
                gmap.optimizer.zero_grad(set_to_none=True)
                gmap.update_learning_rate(self.nr_iters)
                moved = []
                for k in mine:
                    vp = viewpoints[k]
                    po = self._pose_optimizer(vp)
                    if vp.frame_idx != 0:             # the first frame is the gauge: never moved
                        moved.append(po)
                PoseAdam.step_batch(moved)            # the owned keyframes' pose steps in one launch
                for k in mine:
                    self._pose_optimizer(viewpoints[k]).zero_grad()
        return gaussian_split

    # ---- collectives ------------------------------------------------------------------------------------------------
    def _exchange(self, d_norm, d_vis, d_maxr):
        import time
        gmap = self.gmap
        on_gpu = torch.device(gmap.device).type == "cuda"
        if self.time_comm and on_gpu:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        P = len(gmap)
        if self._bucket is None or self._bucket.buf.shape[0] != P or \
                any(a is not b for a, b in zip(self._bucket.params, gmap.params())):
            self._bucket = W.GradBucket(gmap.params(), extra_cols=2)
        _, gn, vs, _ = W.allreduce_window_grads(gmap.params(), d_norm.reshape(-1), d_vis.reshape(-1),
                                                d_maxr, group=self.group, bucket=self._bucket)
        if self.time_comm:
            if on_gpu:
                torch.cuda.synchronize()
            self.exposed_comm_s += time.perf_counter() - t0
        return gn.reshape(-1, 1).clone(), vs.reshape(-1, 1).clone()

    def _union_visible(self, pkgs, mine, P):
        u = torch.zeros(P, dtype=torch.bool, device=self.gmap.device)
        for k in mine:
            u |= pkgs[k]["visibility_filter"]
        if self.world > 1:
            b = u.to(torch.uint8)
            W.all_reduce_(b, op=dist.ReduceOp.MAX, group=self.group)
            u = b.bool()
        return u

    def _prune_covisibility(self, kf_ids):
        """slam_mapper.py:408-448: drop Gaussians of recent keyframes that at most ``prune_coviz`` window keyframes see."""
        gmap = self.gmap
        gmap.nr_obs.zero_()
        for v in self.occ_aware_visibility.values():
            gmap.nr_obs += v.to(torch.int32)
        if not self.first_time_pruned:
            kf_mask = gmap.kf_idx >= 0
            self.first_time_pruned = True
        else:
            kf_mask = gmap.kf_idx >= sorted(kf_ids, reverse=True)[2]
        to_prune = (gmap.nr_obs <= self.prune_coviz) & kf_mask
        gmap.prune_points(to_prune)
        keep = ~to_prune
        self.occ_aware_visibility = {k: v[keep] for k, v in self.occ_aware_visibility.items()}

    def sync_poses(self, viewpoints: Sequence):
        """All-gather the owners' keyframe poses / exposures (before ``push_to_frontend``, slam_mapper.py:553-556)."""
        W.all_gather_poses(viewpoints, self.group)
