"""``optimize_map`` / ``initialize_map`` -- the mapping-window optimisation that drives the rasteriser, optionally sharded by
keyframe over the GPUs of a node (BASELINE.json config C4) and optionally replayed from hipGraphs.

Mirror of ``Mapper.optimize_map`` (/root/reference/utils/slam_mapper.py:244-500) and ``Mapper.initialize_map`` (:169-242):
per iteration render every window keyframe against the shared Gaussians (:273-324), sum ``get_loss_mapping``, ONE backward
(:394), occlusion-aware visibility per keyframe (:400-404), covisibility pruning on request (:408-448), ``max_radii_2d`` +
densification statistics per keyframe (:453-460), ``densify_and_prune`` every ``gaussian_update_every`` iterations
(:462-473), opacity reset of non-visible Gaussians (:476-479), Adam step on the Gaussians, ``update_learning_rate``
(:482-484), Adam on the keyframe poses + ``update_pose`` (:486-496).

ONE implementation of the iteration serves the eager loop, the hipGraph-replayed loop (``use_graph=True``) and the sharded
loop; it has two halves with the collectives between them:

  front   map activations (one launch) -> every owned keyframe: full render (screen-space gradient holder, radii, n_touched),
          fused loss value + gradients, the rasteriser's backward; the gradients of all keyframes meet in ONE flat bucket
          ``[xyz 3 | rgb 3 | opacity 1 | scales 3 | rotation 4 | grad-norm 1 | visible 1] x P`` (``fan_out(out=...)``); one
          statistics launch for all keyframes (``mgs_window_stats``: per-keyframe ||dL/dmean2D|| + visible count + MAX radii
          + packed ``n_touched > 0`` bits)
  [world > 1]  all-reduce(SUM) of the bucket; ONE all-gather carrying MAX radii + visibility bits (MAX taken locally)
  back    statistics folded into the map's, backward of the activations (one launch, no autograd), fused Adam with the
          learning rates in device memory, the xyz schedule + iteration count stepped on the device, the owned keyframes'
          pose steps in one launch.

Nothing in either half synchronises the host, allocates outside the caching allocator or launches a memset / memcpy node, so
both are captured once per (map size, window) and replayed; map surgery (densify / prune / opacity reset) stays eager
between replays, on exactly the iterations the reference does it.  Sharding (`monogs_amd.window`):
rank r renders the window positions ``k % world == r``; the Gaussians are replicated and every rank applies the same Adam
step and the same (identically seeded) surgery to bit-identical inputs, so the replicas stay bit-identical with no
parameter broadcast.  Pose / exposure parameters and their optimiser state live on the owning rank only; ``sync_poses``
all-gathers them before the map is handed on (:553-556).
"""
from __future__ import annotations

import contextlib
import math
import time
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib, camera as cam, fused_losses, rasterizer as _rast, window as W
from .gaussian_map import GaussianMap
from .gaussian_optim import activate, fan_out, window_stats
from .pose_optim import PoseAdam
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, _device_guard, _stream
from .renderer import render

BUCKET_COLS = 16          # floats per Gaussian in the exchange bucket (see the module docstring)
_GRAD_COLS = 14


def render_map(vp, intr, gmap: GaussianMap, bg):
    """``render()`` with the map's activations (normalize / exp / sigmoid, forward and backward) in one launch each."""
    if gmap.fused_adam and gmap._rotation.requires_grad:
        rot, scales3, opac = activate(gmap._rotation, gmap._scaling, gmap._opacity)
        return render(vp, intr, gmap.get_xyz, rot, scales3, opac, gmap.get_features, bg)
    return render(vp, intr, gmap.get_xyz, gmap.get_rotation, gmap.get_scaling, gmap.get_opacity, gmap.get_features, bg)


class _Plan:
    """Static state of the window iteration for one (map size, window, loss flavour): everything a captured half points at."""

    def __init__(self, mapper: "WindowMapper", viewpoints: Sequence, init: bool):
        gmap, dev = mapper.gmap, mapper.gmap.device
        self.P, self.n, self.init = len(gmap), len(viewpoints), bool(init)
        self.mine = mapper.owned(self.n)
        self.vps = [viewpoints[k] for k in self.mine]
        self.key = (self.P, tuple(id(v) for v in viewpoints), self.init, int(mapper.intr.height), int(mapper.intr.width),
                    tuple(id(p) for p in gmap.params()))
        P, f32 = self.P, dict(dtype=torch.float32, device=dev)
        self.sd = int(gmap._scaling.shape[1])
        self.bucket = torch.zeros(P * BUCKET_COLS, **f32)
        self.rot, self.scales3, self.opac = torch.empty(P, 4, **f32), torch.empty(P, 3, **f32), torch.empty(P, 1, **f32)
        self.d_rot, self.d_scale, self.d_opac = torch.empty(P, 4, **f32), torch.empty(P, self.sd, **f32), torch.empty(P, 1, **f32)
        # the screen-space gradient holders of render() (zeros whose .grad receives dL/dmean2D), one per owned keyframe,
        # allocated once: the rasteriser never reads their values
        self.holders = [torch.zeros(P, 3, requires_grad=True, **f32) for _ in self.mine]
        self.words = (P + 63) // 64
        self.rows = W.rows_per_rank(self.n, mapper.world)
        # what travels in the all-gather: MAX radii of this rank's keyframes, then `rows` rows of visibility bits
        self.g_words = (P + 1) // 2 + self.rows * self.words                  # int64 words: P floats (padded) + the bit rows
        self.gin = torch.zeros(self.g_words, dtype=torch.int64, device=dev)
        self.gout = torch.zeros(mapper.world, self.g_words, dtype=torch.int64, device=dev) if mapper.sharded else None
        self.maxr = self.gin[:(P + 1) // 2].view(torch.float32)[:P]
        self.bits = self.gin[(P + 1) // 2:].view(self.rows, self.words)
        self.lgs: List = []
        self.radii: List = []
        self.n_touched: List = []
        self.graphs = None              # (front, back) or (whole,) once captured
        self.iter_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        # pipelined exchange (WindowMapper.exchange = "per_keyframe"): one gradient bucket per owned keyframe, reduced on
        # its own while the next keyframe renders; `slot_graphs`: one captured (render, loss, backward) per owned keyframe
        # (decided from rank-INDEPENDENT quantities: every rank issues exactly `rows` slot collectives of the same size, a rank
        #  that owns fewer keyframes -- any window that is not a multiple of the world size -- contributes zeroed slots)
        self.pipelined = bool(mapper.sharded and mapper.exchange == "per_keyframe" and self.rows > 1)
        self.slot_flat = [torch.zeros(P * _GRAD_COLS, **f32) for _ in range(self.rows)] if self.pipelined else []
        self.slot_graphs = None

    def grad_view(self, col0: int, cols: int):
        P = self.P
        return self.bucket[P * col0:P * (col0 + cols)].view(P, cols)


class WindowMapper:
    # the values hard-coded in the fork's mapper (/root/reference/utils/slam_mapper.py:64-89)
    init_itr_num = 1050
    init_gaussian_update = 100
    init_gaussian_reset = 500
    init_gaussian_th = 0.005
    init_gaussian_extent = 30.0
    gaussian_update_every = 150
    gaussian_update_offset = 50
    gaussian_th = 0.7
    gaussian_extent = 1.0
    gaussian_reset = 2001
    size_threshold = 20
    densify_grad_threshold = 0.0002      # /root/reference/configs/mono/tum/base_config.yaml:66
    densify_from_iter = 500              # :63
    prune_coviz = 3

    def __init__(self, gmap: GaussianMap, intr, bg, group=None, window_size: int = 8, seed: int = 0,
                 lr_rot: float = 0.003 * 0.5, lr_trans: float = 0.001 * 0.5, lr_exposure: float = 0.01,
                 use_graph: bool = False, force_collectives: bool = False):
        assert gmap.fused_adam, "WindowMapper drives the fused optimiser (GaussianAdam)"
        self.gmap, self.intr, self.bg, self.group = gmap, intr, bg, group
        self.window_size, self.seed = int(window_size), int(seed)
        self.world, self.rank = W._world(group), W._rank(group)
        self.lrs = (lr_rot, lr_trans, lr_exposure)
        self.use_graph = bool(use_graph)
        # the collectives run when there is more than one rank -- or on request in a one-rank group (a test drives the RCCL
        # code path of every exchange that way on a one-GPU box)
        self.sharded = self.world > 1 or (bool(force_collectives) and dist.is_available() and dist.is_initialized())
        self.nr_iters = 0
        self.first_time_pruned = False
        self.occ_aware_visibility: Dict[int, torch.Tensor] = {}     # kf id -> bool[P]
        self._pose_opt: Dict[int, PoseAdam] = {}                    # id(viewpoint) -> optimiser state (owning rank)
        self.exposed_comm_s = 0.0       # wall time spent waiting in collectives (diagnostic, synchronises when on)
        self.time_comm = False
        self.map_surgery = True          # False: no densify_and_prune / opacity reset (fixed-size workloads: benchmarks, tests)
        self.keep_reduced_grads = False  # tests: clones of the (all-reduced) Gaussian gradients of the last iteration
        self.parallel_keyframes = None   # render / back-propagate the owned keyframes on a stream each; None: when captured
                                         # (pays inside a replay: 863 -> 1254 it/s; an eager loop is bound by the host anyway)
        self.min_graph_iters = 8         # shorter runs are not worth a capture
        self.max_replays_per_capture = 256   # then an eager iteration refreshes the instance capacities and the plan is re-captured
        self._streams: List = []
        self.last_grads = None
        self._plan: Optional[_Plan] = None
        self._carry = None               # gradients a pruning call left behind (see optimize_map)
        self._pool = None
        self.stats = dict(captures=0, replays=0, eager_iters=0, capture_s=0.0, replay_s=0.0, replay_kf=0)
        self.time_replays = False        # measure the replay chunks (one synchronisation at either end of a chunk)
        self.coviz_log: List = []        # (map size, Gaussians dropped) of every covisibility prune
        # How the Gaussian gradients cross the ranks (sharded only).
        #   "bucket" (default): every rank sums its keyframes' gradients locally and ONE all-reduce follows the last backward --
        #       the fewest bytes, nothing to overlap with.
        #   "per_keyframe": with several keyframes per rank (window / world > 1) each owned keyframe's gradients are reduced on
        #       their own, issued as soon as that keyframe's backward is queued, so the collective of keyframe k runs (on RCCL's
        #       stream) while keyframe k + 1 renders; the reduced buckets are then added in keyframe order.  window / world times
        #       the bytes: pays when a bucket's all-reduce takes about as long as a render (maps of millions of Gaussians), not
        #       at SLAM sizes.  `overlap_exchange = False` issues the same collectives after the last backward instead: the same
        #       arithmetic in the same order, hence bit-identical results -- the reference the overlapped schedule is tested against.
        self.exchange = "bucket"
        self.overlap_exchange = True

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

    @property
    def last_loss(self):
        """Sum of the owned keyframes' loss values of the last iteration (device scalar), or None."""
        p = self._plan
        if p is None or not p.lgs:
            return None
        return torch.stack([lg.loss for lg in p.lgs]).sum()

    # ---- plans and graphs ---------------------------------------------------------------------------------------------
    def _get_plan(self, viewpoints, init) -> _Plan:
        p = self._plan
        key = (len(self.gmap), tuple(id(v) for v in viewpoints), bool(init), int(self.intr.height), int(self.intr.width),
               tuple(id(q) for q in self.gmap.params()))
        if p is None or p.key != key:
            self._drop_plan()
            p = self._plan = _Plan(self, viewpoints, init)
        return p

    def _drop_plan(self):
        if self._plan is not None and self._plan.graphs is not None:
            _rast.clear_graph_flags()
        self._plan = None

    # ---- the two halves of an iteration -----------------------------------------------------------------------------
    def _activations(self, p: _Plan):
        gmap, lib, P = self.gmap, _lib.load(), p.P
        with _device_guard(gmap._xyz.device):
            _lib.check(lib.mgs_activate_forward(P, p.sd, gmap._rotation.data_ptr(), gmap._scaling.data_ptr(),
                                                gmap._opacity.data_ptr(), p.rot.data_ptr(), p.scales3.data_ptr(),
                                                p.opac.data_ptr(), _stream()), "mgs_activate_forward")

    def _cut(self, p: _Plan):
        """The five tensors the renders differentiate with respect to: leaves that CUT the graph at the activations (their
        backward is one explicit launch in `_back`, after the collectives)."""
        gmap = self.gmap
        return [t.detach().requires_grad_(True) for t in (gmap._xyz, gmap._rgb, p.opac, p.scales3, p.rot)]

    def _render_slot(self, p: _Plan, j: int, fan):
        """Forward of owned keyframe j: the full render + the fused loss value and gradients.  Returns what its backward needs."""
        vp = p.vps[j]
        xyz_k, feat_k, opac_k, sc_k, rot_k = fan
        h = p.holders[j]
        h.grad = None
        color, radii, depth, n_touched = self._rasterize(vp, xyz_k, rot_k, sc_k, opac_k, feat_k, h)
        lg = fused_losses.loss_grads(color, depth, None, vp, tracking=False, init=p.init)
        p.lgs.append(lg)
        p.radii.append(radii)
        p.n_touched.append(n_touched)
        return [color, depth], [lg.d_render, lg.d_depth]

    @staticmethod
    def _take_exposure_grads(vp, lg):
        if lg.has_exposure:         # (added to what a pruning call left there, like autograd's accumulation)
            for q, g in ((vp.exposure_a, lg.d_exposure_a), (vp.exposure_b, lg.d_exposure_b)):
                q.grad = g if q.grad is None else q.grad + g

    def _stats(self, p: _Plan, accumulate_stats: bool):
        """Per-keyframe statistics, one launch (per-keyframe norm BEFORE any summation)."""
        gmap = self.gmap
        if accumulate_stats:            # single rank: straight into the map's running statistics, in keyframe order
            norm, vis, maxr = gmap.xyz_gradient_accum, gmap.denom, gmap.max_radii_2d
        else:
            norm, vis, maxr = p.grad_view(14, 1), p.grad_view(15, 1), p.maxr
        window_stats([h.grad for h in p.holders], p.radii, p.n_touched, norm, vis, maxr, accumulate_stats, p.bits)

    def _front(self, p: _Plan, accumulate_stats: bool, parallel: bool):
        """Renders + losses + ONE backward of the owned keyframes into the bucket, then the statistics launch."""
        P = p.P
        self._activations(p)
        cut = self._cut(p)
        p.lgs, p.radii, p.n_touched = [], [], []
        if not p.mine:
            p.bucket[:P * _GRAD_COLS].zero_()
        else:
            fans = fan_out(len(p.mine), *cut, out=p.bucket[:P * _GRAD_COLS])
            streams = self._kf_streams(len(p.mine)) if (parallel and len(p.mine) > 1) else None
            main = torch.cuda.current_stream() if streams else None
            outs, grads = [], []
            for j in range(len(p.vps)):
                if streams:
                    streams[j].wait_stream(main)
                with (torch.cuda.stream(streams[j]) if streams else contextlib.nullcontext()):
                    o, g = self._render_slot(p, j, fans[j])
                outs += o
                grads += g
            # (no join before the backward: every keyframe's backward runs on the stream of its forward, behind it, and the
            #  node that adds the gradients up waits for all of them)
            torch.autograd.backward(outs, grads)
            if streams:
                for st in streams:
                    main.wait_stream(st)
            for vp, lg in zip(p.vps, p.lgs):
                self._take_exposure_grads(vp, lg)
        self._stats(p, accumulate_stats)

    # ---- pipelined exchange: one bucket per owned keyframe, its all-reduce behind the next keyframe's render ------------
    def _slot(self, p: _Plan, j: int):
        """Render + loss + backward of owned keyframe j alone; its map gradients land in ``p.slot_flat[j]`` (bucket layout)."""
        fan = fan_out(1, *self._cut(p), out=p.slot_flat[j])[0]
        o, g = self._render_slot(p, j, fan)
        torch.autograd.backward(o, g)
        self._take_exposure_grads(p.vps[j], p.lgs[-1])

    def _front_pipelined(self, p: _Plan):
        """The front half with ``exchange = "per_keyframe"``: activations, then per owned keyframe (render, loss, backward) and
        the all-reduce of ITS bucket -- asynchronous, so that it runs behind the next keyframe's kernels -- then the reduced
        buckets added up in keyframe order and the statistics launch.  The two statistics columns and the all-gather follow in
        `_exchange`.  Captured plans replay one graph per keyframe (`p.slot_graphs`) and one for the tail."""
        if p.slot_graphs is None:
            self._activations(p)
            p.lgs, p.radii, p.n_touched = [], [], []
            produce = lambda j: self._slot(p, j)                    # noqa: E731
        else:
            p.slot_graphs[0].replay()                                # (the activations)
            produce = lambda j: p.slot_graphs[1 + j].replay()        # noqa: E731
        W.pipelined_all_reduce(p.rows, produce, p.slot_flat, group=self.group, overlap=self.overlap_exchange, n_owned=len(p.mine))
        if p.slot_graphs is None:
            self._sum_slots(p)
        else:
            p.slot_graphs[-1].replay()

    def _sum_slots(self, p: _Plan):
        from .gaussian_optim import sum_buffers
        sum_buffers(p.slot_flat, out=p.bucket[:p.P * _GRAD_COLS])          # fixed order: keyframe 0, 1, ... of this rank
        self._stats(p, accumulate_stats=False)

    def _rasterize(self, vp, xyz, rot, scales3, opac, feat, holder):
        """The rasteriser call of ``render()`` (/root/reference/gaussian_splatting/gaussian_renderer/__init__.py:52-156)."""
        intr = self.intr
        view, full, campos = cam.cached_camera_tensors(vp, vp.R, vp.T, intr.projection_matrix)
        rs = GaussianRasterizationSettings(
            image_height=int(intr.height), image_width=int(intr.width),
            tanfovx=math.tan(intr.FoVx * 0.5), tanfovy=math.tan(intr.FoVy * 0.5), bg=self.bg, scale_modifier=1.0,
            viewmatrix=view, projmatrix=full, projmatrix_raw=intr.projection_matrix, sh_degree=0, campos=campos,
            prefiltered=False, debug=False)
        color, radii, depth, _, n_touched = GaussianRasterizer(rs)(
            means3D=xyz, means2D=holder, opacities=opac, colors_precomp=feat, scales=scales3, rotations=rot,
            theta=vp.cam_rot_delta, rho=vp.cam_trans_delta)
        return color, radii, depth, n_touched

    def _exchange(self, p: _Plan, grads: bool = True, reduced: bool = False):
        """The collectives of one iteration: SUM of the bucket, one all-gather of MAX radii + visibility bits.
        ``reduced``: the gradient columns were reduced keyframe by keyframe already (`_front_pipelined`); the two
        statistics columns remain."""
        if not self.sharded:
            return
        on_gpu = torch.device(self.gmap.device).type == "cuda"
        if self.time_comm and on_gpu:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if grads:
            # (pipelined plans reduced the gradient columns keyframe by keyframe already: the two statistics columns remain)
            W.all_reduce_(p.bucket[p.P * _GRAD_COLS:] if reduced else p.bucket, group=self.group)
        W.all_gather_into_(p.gout.view(-1), p.gin, group=self.group)
        if self.time_comm:
            if on_gpu:
                torch.cuda.synchronize()
            self.exposed_comm_s += time.perf_counter() - t0

    def _back(self, p: _Plan, viewpoints, pose_steps: bool, lr_update: bool, surgery=None):
        """Statistics into the map, backward of the activations, (eager only: map surgery,) Adam, schedule, pose steps."""
        gmap, lib, P = self.gmap, _lib.load(), p.P
        with _device_guard(gmap._xyz.device):
            if self.sharded:
                maxr = torch.amax(p.gout[:, :(P + 1) // 2].view(torch.float32)[:, :P], dim=0)
                _lib.check(lib.mgs_window_apply(P, p.grad_view(14, 1).data_ptr(), p.grad_view(15, 1).data_ptr(),
                                                maxr.data_ptr(), gmap.xyz_gradient_accum.data_ptr(), gmap.denom.data_ptr(),
                                                gmap.max_radii_2d.data_ptr(), _stream()), "mgs_window_apply")
            _lib.check(lib.mgs_activate_backward(P, p.sd, gmap._rotation.data_ptr(), p.scales3.data_ptr(), p.opac.data_ptr(),
                                                 p.grad_view(10, 4).data_ptr(), p.grad_view(7, 3).data_ptr(),
                                                 p.grad_view(6, 1).data_ptr(), p.d_rot.data_ptr(), p.d_scale.data_ptr(),
                                                 p.d_opac.data_ptr(), _stream()), "mgs_activate_backward")
        # the leaves' gradients, exactly where autograd would have put them -- BEFORE any surgery: densify_and_prune replaces
        # every tensor (new leaves, no gradient: no step, as torch.optim.Adam), an opacity reset replaces the opacities only
        for q, g in zip(gmap.params(), (p.grad_view(0, 3), p.grad_view(3, 3), p.d_opac, p.d_scale, p.d_rot)):
            q.grad = g
        gaussian_split = bool(surgery(p)) if surgery is not None else False
        if self.keep_reduced_grads:
            self.last_grads = [None if q.grad is None else q.grad.clone() for q in gmap.params()]
        gmap.optimizer.step()                     # (after surgery every .grad is None: no step, as torch.optim.Adam)
        if lr_update and gmap.lr_schedule is not None:
            s = gmap.lr_schedule
            with _device_guard(gmap._xyz.device):
                _lib.check(lib.mgs_lr_schedule_step(p.iter_dev.data_ptr(), gmap.optimizer.device_lrs().data_ptr(),
                                                    float(s["lr_init"]), float(s["lr_final"]), int(s.get("lr_delay_steps", 0)),
                                                    float(s.get("lr_delay_mult", 1.0)), int(s["max_steps"]), _stream()),
                           "mgs_lr_schedule_step")
        if pose_steps:
            moved = [self._pose_optimizer(vp) for vp in p.vps if vp.frame_idx != 0]      # the first frame is the gauge
            PoseAdam.step_batch(moved)                                                   # one launch
        return gaussian_split

    def _zero_grads(self, p: Optional[_Plan], viewpoints):
        """optimizer.zero_grad(set_to_none=True) + keyframe_optimizers.zero_grad(set_to_none=True) (slam_mapper.py:483,487)."""
        self.gmap.optimizer.zero_grad(set_to_none=True)
        for k in self.owned(len(viewpoints)):
            self._pose_optimizer(viewpoints[k]).zero_grad()
        if p is not None:
            for h in p.holders:
                h.grad = None

    # ---- capture ----------------------------------------------------------------------------------------------------
    def _capture(self, p: _Plan, viewpoints, pose_steps, lr_update):
        t0 = time.perf_counter()
        self._zero_grads(p, viewpoints)
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        par = True if self.parallel_keyframes is None else bool(self.parallel_keyframes)
        single = not self.sharded
        if p.pipelined:
            # activations | one graph per owned keyframe | slot sum + statistics; the collectives run between them
            p.lgs, p.radii, p.n_touched = [], [], []
            graphs = []
            for body in [lambda: self._activations(p)] + [(lambda j=j: self._slot(p, j)) for j in range(len(p.mine))] + \
                    [lambda: self._sum_slots(p)]:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self._pool):
                    body()
                graphs.append(g)
            p.slot_graphs = graphs
            g1 = None
        else:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, pool=self._pool):
                self._front(p, accumulate_stats=single, parallel=par)
                if single:
                    self._back(p, viewpoints, pose_steps, lr_update)
        if single:
            p.graphs = (g1,)
        else:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=self._pool):
                self._back(p, viewpoints, pose_steps, lr_update)
            p.graphs = (g1, g2)
        self.stats["captures"] += 1
        self.stats["capture_s"] += time.perf_counter() - t0

    def _replay(self, p: _Plan, n: int):
        if self.time_replays:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if len(p.graphs) == 1:
            for _ in range(n):
                p.graphs[0].replay()
        else:
            for _ in range(n):
                if p.pipelined:
                    self._front_pipelined(p)
                else:
                    p.graphs[0].replay()
                self._exchange(p, reduced=p.pipelined)
                p.graphs[1].replay()
        if self.time_replays:
            torch.cuda.synchronize()
            self.stats["replay_s"] += time.perf_counter() - t0
            self.stats["replay_kf"] += n * p.n
        self.stats["replays"] += n

    # ---- the loop shared by optimize_map and initialize_map -----------------------------------------------------------
    def _sync_schedule(self, p: _Plan):
        """Device copies of the mapper's iteration count and of the learning rates (what the captured schedule steps)."""
        p.iter_dev.fill_(self.nr_iters)
        self.gmap.optimizer.device_lrs()

    def _run(self, viewpoints, kf_ids, iters: int, init: bool, pose_steps: bool, lr_update: bool, surgery_at) -> bool:
        gmap = self.gmap
        gaussian_split = False
        done = 0
        on_gpu = torch.device(gmap.device).type == "cuda"
        while done < iters:
            p = self._get_plan(viewpoints, init)
            self._sync_schedule(p)
            # iterations until (and including) the next one that does map surgery
            ahead = 0
            while done + ahead < iters and surgery_at(self.nr_iters + ahead + 1, done + ahead) is None:
                ahead += 1
            plain = ahead                               # iterations without surgery from here
            if plain > 0:
                graph_ok = self.use_graph and on_gpu and (plain >= self.min_graph_iters or p.graphs is not None)
                # The first iteration of a new plan is always eager: it records the capacity hints of this map size (one host
                # read-back per render) and is what a capture needs to have run before it.  So is the one that has to take in
                # what a pruning call left behind (Gaussian gradients in `_carry`, pose gradients in the parameters' .grad).
                pending = self._carry is not None or any(
                    q.grad is not None for vp in p.vps for q in (vp.cam_rot_delta, vp.cam_trans_delta, vp.exposure_a, vp.exposure_b))
                n_left = plain
                if p.graphs is None or pending or not graph_ok:
                    self._iterate_eager(p, viewpoints, pose_steps, lr_update, None, parallel=False)
                    n_left -= 1
                if n_left > 0 and graph_ok:
                    # A captured iteration renders with the instance capacity its eager predecessor recorded (x 1.5).  The
                    # reference's surgery forces a new plan every <= 150 iterations; a run WITHOUT it (1 049 replays of one
                    # initialisation iteration) lets the splats grow past that capacity.  So a capture serves at most
                    # `max_replays_per_capture` replays: then one eager iteration records fresh capacities and the plan is captured
                    # again (~10 ms per 256 iterations); the flag is read after every chunk, so an overflow is reported within a
                    # chunk of where it happened instead of at the end of the run.
                    while n_left > 0:
                        if p.graphs is None:
                            self._capture(p, viewpoints, pose_steps, lr_update)
                        n = min(n_left, self.max_replays_per_capture)
                        self._replay(p, n)
                        self.nr_iters += n
                        n_left -= n
                        if _rast.check_overflow():
                            raise RuntimeError("binning capacity overflow inside the captured mapping iteration")
                        if n_left > 0:
                            p.graphs = p.slot_graphs = None
                            self._sync_schedule(p)
                            self._iterate_eager(p, viewpoints, pose_steps, lr_update, None, parallel=False)
                            n_left -= 1
                            self._sync_schedule(p)
                else:
                    for _ in range(n_left):
                        self._iterate_eager(p, viewpoints, pose_steps, lr_update, None, parallel=False)
                done += plain
                self._zero_grads(p, viewpoints)
            if done < iters:                            # the iteration with map surgery in it, eager
                fn = surgery_at(self.nr_iters + 1, done)
                gaussian_split |= self._iterate_eager(p, viewpoints, pose_steps, lr_update, fn, parallel=False)
                self._zero_grads(None, viewpoints)
                done += 1
        self._materialise_visibility(kf_ids)
        if gmap.optimizer.lr_dev is not None and gmap.lr_schedule is not None:
            gmap.optimizer.sync_lrs_from_device()
        return gaussian_split

    def _iterate_eager(self, p: _Plan, viewpoints, pose_steps, lr_update, surgery, parallel) -> bool:
        self.nr_iters += 1
        self.stats["eager_iters"] += 1
        reduced = p.pipelined and self._carry is None
        if reduced:
            self._front_pipelined(p)
        else:       # (an iteration that takes in a pruning call's carried gradients uses the plain bucket: they join it before the all-reduce)
            self._front(p, accumulate_stats=not self.sharded, parallel=parallel)
        if self._carry is not None:
            if self._carry.shape[0] == p.P * _GRAD_COLS:
                p.bucket[:p.P * _GRAD_COLS] += self._carry
            self._carry = None
        self._exchange(p, reduced=reduced)
        split = self._back(p, viewpoints, pose_steps, lr_update, surgery)
        self._zero_grads(p if not split else None, viewpoints)
        return split

    def _materialise_visibility(self, kf_ids):
        """``occ_aware_visibility[kf] = n_touched_kf > 0`` of the LAST iteration for every keyframe of the window
        (/root/reference/utils/slam_mapper.py:400-404; the reference rebuilds the dict every iteration, only the last
        survives)."""
        p = self._plan
        if p is None:
            return
        out = {}
        for k in range(p.n):
            src = p.bits if not self.sharded else p.gout[k % self.world, (p.P + 1) // 2:].view(p.rows, p.words)
            words = src[k // self.world]
            out[kf_ids[k]] = W.unpack_bits(words.view(torch.uint8), p.P)
        self.occ_aware_visibility = out

    # ---- one call = Mapper.optimize_map(cur_kf_list, prune, iters) ------------------------------------------------------
    def optimize_map(self, viewpoints: Sequence, kf_ids: Optional[Sequence[int]] = None, prune: bool = False,
                     iters: int = 1, init: bool = False) -> bool:
        """``viewpoints``: the window's keyframes (every rank holds all of them; only the owned ones are rendered).
        ``kf_ids``: their keyframe ids (default: ``vp.frame_idx``).  Returns ``gaussian_split`` as the reference."""
        n = len(viewpoints)
        if n == 0:
            return False
        kf_ids = [int(v.frame_idx) for v in viewpoints] if kf_ids is None else [int(k) for k in kf_ids]
        if prune:
            for _ in range(iters):
                self._prune_call(viewpoints, kf_ids, init)
            return False

        def surgery_at(nr_iters, _i):
            if not self.map_surgery:
                return None
            update = nr_iters % self.gaussian_update_every == self.gaussian_update_offset
            reset = (nr_iters % self.gaussian_reset) == 0 and not update
            if not (update or reset):
                return None

            def fn(p):
                if update:
                    self.gmap.densify_and_prune(self.densify_grad_threshold, self.gaussian_th, self.gaussian_extent,
                                                self.size_threshold,
                                                generator=W.split_generator(self.gmap.device, self.seed, nr_iters))
                else:       # every keyframe's visibility_filter (radii > 0); only their union matters
                    self.gmap.reset_opacity_nonvisible([self._union_visible(p)])
                return True
            return fn
        return self._run(viewpoints, kf_ids, iters, init, pose_steps=True, lr_update=True, surgery_at=surgery_at)

    def initialize_map(self, viewpoint, kf_id: Optional[int] = None, iters: Optional[int] = None) -> None:
        """``Mapper.initialize_map`` (/root/reference/utils/slam_mapper.py:169-242): ``init_itr_num`` single-camera
        iterations with the ``init`` loss, densify_and_prune every ``init_gaussian_update`` iterations (the first one
        included, no screen-size limit), ``reset_opacity`` when the iteration count reaches ``init_gaussian_reset`` or
        ``densify_from_iter``, no learning-rate schedule, no pose step."""
        iters = self.init_itr_num if iters is None else int(iters)
        kf_id = int(viewpoint.frame_idx) if kf_id is None else int(kf_id)

        def surgery_at(nr_iters, i):
            if not self.map_surgery:
                return None
            update = i % self.init_gaussian_update == 0
            reset = nr_iters == self.init_gaussian_reset or nr_iters == self.densify_from_iter
            if not (update or reset):
                return None

            def fn(p):
                if update:
                    self.gmap.densify_and_prune(self.densify_grad_threshold, self.init_gaussian_th, self.init_gaussian_extent,
                                                None, generator=W.split_generator(self.gmap.device, self.seed, nr_iters))
                if reset:
                    self.gmap.reset_opacity()
                return True
            return fn
        self._run([viewpoint], [kf_id], iters, init=True, pose_steps=False, lr_update=False, surgery_at=surgery_at)

    # ---- the pruning call (prune=True, one iteration, no optimiser step) --------------------------------------------
    def _prune_call(self, viewpoints, kf_ids, init):
        """slam_mapper.py:394-451: render + backward as always, visibility, covisibility pruning when the window is full,
        return before the statistics and the optimiser steps.  When the window is not full yet the gradients of this
        iteration STAY in ``.grad`` (nothing zeroes them) and the next call's backward adds to them: they are kept here
        at the level they are produced (activated tensors, this rank's keyframes only) and join the bucket of the next
        iteration BEFORE its all-reduce -- so that the sum over ranks counts them once."""
        self.nr_iters += 1
        self.stats["eager_iters"] += 1
        p = self._get_plan(viewpoints, init)
        # (statistics launch: the bits only; its three sums land in the bucket's statistics columns and are ignored)
        self._front(p, accumulate_stats=False, parallel=False)
        self._exchange(p, grads=False)
        self._materialise_visibility(kf_ids)
        if len(viewpoints) == self.window_size:
            # prune_points replaces every Gaussian tensor by a new leaf, so the Gaussian gradients of this call vanish with the
            # old ones; nothing touches the keyframes' pose / exposure .grad, which the reference returns with (no zero_grad on
            # this path, slam_mapper.py:408-451): the next call's first backward adds to them, as in the other branch
            self._carry = None
            self.gmap.optimizer.zero_grad(set_to_none=True)
            for h in p.holders:
                h.grad = None
            self._prune_covisibility(kf_ids)
        else:
            # (the pose / exposure gradients of the owned keyframes stay in their .grad, as in the reference; the next
            #  backward adds to them)
            carry = p.bucket[:p.P * _GRAD_COLS].clone()
            self._carry = carry if (self._carry is None or self._carry.shape != carry.shape) else self._carry + carry
            for h in p.holders:
                h.grad = None

    def _union_visible(self, p: _Plan):
        u = torch.zeros(p.P, dtype=torch.bool, device=self.gmap.device)
        for r in p.radii:
            u |= r > 0
        if self.sharded:
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
        n0 = len(gmap)
        gmap.prune_points(to_prune)
        self.coviz_log.append((n0, n0 - len(gmap)))
        keep = ~to_prune
        self.occ_aware_visibility = {k: v[keep] for k, v in self.occ_aware_visibility.items()}
        self._drop_plan()

    def sync_poses(self, viewpoints: Sequence):
        """All-gather the owners' keyframe poses / exposures (before ``push_to_frontend``, slam_mapper.py:553-556)."""
        W.all_gather_poses(viewpoints, self.group, force=self.sharded)
