"""A minimal tracking + mapping loop around ``render()`` for measuring the SLAM-level metric
(tracking / mapping FPS, BASELINE.json configs 3-4) without the reference's control plane.

This is measurement scaffolding on the CALLER's side of the boundary, not a re-implementation of
MonoGS: no viewer, dataset parsers, keyframe selection by overlap (``run_slam_two_process`` has the
reference's tracker / mapper process split).  It reproduces the two hot loops that drive the rasteriser
exactly as the reference does:

* tracking  (/root/reference/utils/slam_tracker.py:83-193): pose-only Adam (rot 0.003, trans 0.001,
  exposure 0.01 -- /root/reference/configs/mono/tum/base_config.yaml:46-48), <= ``tracking_itr_num``
  iterations of render -> get_loss_tracking -> backward -> step -> update_pose, early exit when the
  retraction step is < 1e-4;
* mapping   (/root/reference/utils/slam_mapper.py:169-242,244-500) through ``monogs_amd.mapping.WindowMapper``: per
  iteration render every window keyframe, sum get_loss_mapping, ONE backward, statistics, Adam step on the Gaussians and
  on the window poses; with ``map_surgery`` the reference's densify_and_prune / opacity resets / covisibility pruning on
  its own schedule.

The camera objects are duck-typed stand-ins for the reference's CameraIntrinsics / CameraExtrinsics
(/root/reference/utils/camera_utils.py:8-79,82-221).  The datasets of configs 3-4 are not available offline; frames come
from an opaque box room ray-cast analytically (``make_room_sequence``: what bench.py runs, survives the reference's
pruning) or, historically, from a seeded cloud of Gaussians rendered by the same rasteriser (``make_sequence``).
"""
from __future__ import annotations

import math
import time
from typing import List

import torch

from . import camera as cam
from .renderer import render
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from . import fused_losses
from .gaussian_map import GaussianMap
from .gaussian_optim import activate
from .pose_optim import PoseAdam
from .synthetic import make_scene


class Intrinsics:
    def __init__(self, k: dict, device):
        self.k, self.device = dict(k), device
        self.height, self.width = k["H"], k["W"]
        self.fx, self.fy, self.cx, self.cy = k["fx"], k["fy"], k["cx"], k["cy"]
        m = cam.camera_matrices(torch.eye(3), torch.zeros(3), k["fx"], k["fy"], k["cx"], k["cy"], k["W"], k["H"])
        self.projection_matrix = m.projmatrix_raw.to(device)        # transposed, as the reference's property
        self.FoVx, self.FoVy = 2 * math.atan(m.tanfovx), 2 * math.atan(m.tanfovy)


def scharr_grad_mask(rgb: torch.Tensor, edge_threshold: float = 1.1, eps: float = 0.01) -> torch.Tensor:
    """``CameraExtrinsics.compute_grad_mask`` (/root/reference/utils/camera_utils.py:185-216): Scharr gradient of the grey
    image (``image_gradient``, utils/slam_utils.py:6-23: reflect padding, normalised by 32), zeroed where a 3x3
    neighbourhood holds a pixel <= ``eps`` (``image_gradient_mask``, :26-40), thresholded at ``edge_threshold`` x median."""
    gray = rgb.mean(dim=0, keepdim=True)
    kx = torch.tensor([[3.0, 10.0, 3.0], [0.0, 0.0, 0.0], [-3.0, -10.0, -3.0]], device=rgb.device)
    ky = torch.tensor([[3.0, 0.0, -3.0], [10.0, 0.0, -10.0], [3.0, 0.0, -3.0]], device=rgb.device)
    pad = torch.nn.functional.pad(gray[None], (1, 1, 1, 1), mode="reflect")
    conv = torch.nn.functional.conv2d
    gv = conv(pad, kx.view(1, 1, 3, 3))[0] / 32.0
    gh = conv(pad, ky.view(1, 1, 3, 3))[0] / 32.0
    full = conv((pad.abs() > eps).float(), torch.ones(1, 1, 3, 3, device=rgb.device))[0] == 9.0
    mag = torch.sqrt((gv * full) ** 2 + (gh * full) ** 2)[0]
    return mag > mag.median() * edge_threshold


class Viewpoint:
    def __init__(self, idx, rgb, depth, device, gt_R=None, gt_T=None):
        self.frame_idx, self.device = idx, device
        self.R = torch.eye(3, device=device)
        self.T = torch.zeros(3, device=device)
        self.R_gt, self.T_gt = gt_R, gt_T
        self.rgb, self.depth = rgb, depth
        self.mask = torch.ones_like(depth, dtype=torch.bool)
        self.grad_mask = scharr_grad_mask(rgb)
        z = lambda n, v=0.0: torch.nn.Parameter(torch.full((n,), v, device=device))  # noqa: E731
        self.cam_rot_delta, self.cam_trans_delta = z(3), z(3)
        self.exposure_a, self.exposure_b = z(1), z(1)

    @property
    def world_view_transform(self):
        return cam.world2view(self.R, self.T).transpose(0, 1)

    @property
    def camera_center(self):
        return self.world_view_transform.inverse()[3, :3]

    def update_RT(self, R, t):
        self.R, self.T = R.to(self.device).contiguous(), t.to(self.device).contiguous()

    def retract(self, thr=1e-4) -> bool:
        """update_pose (/root/reference/utils/pose_utils.py:76-93) on device tensors."""
        tau = torch.cat([self.cam_trans_delta.data, self.cam_rot_delta.data])
        Tm = torch.eye(4, device=self.device)
        Tm[:3, :3], Tm[:3, 3] = self.R, self.T
        Tn = cam.se3_exp(tau) @ Tm
        self.R, self.T = Tn[:3, :3], Tn[:3, 3]
        conv = bool(tau.norm() < thr)
        self.cam_rot_delta.data.zero_()
        self.cam_trans_delta.data.zero_()
        return conv


def _render(vp, intr, gmap: GaussianMap, bg):
    if gmap.fused_adam and gmap._rotation.requires_grad:      # one launch for normalize / exp / sigmoid (+ backward)
        rot, scales3, opac = activate(gmap._rotation, gmap._scaling, gmap._opacity)
        return render(vp, intr, gmap.get_xyz, rot, scales3, opac, gmap.get_features, bg)
    return render(vp, intr, gmap.get_xyz, gmap.get_rotation, gmap.get_scaling, gmap.get_opacity, gmap.get_features, bg)


class TrackingGraph:
    """The tracking iteration (render -> fused loss -> backward -> fused pose step) captured ONCE per map version
    in a hipGraph and replayed for every iteration of every frame tracked against that map.

    * The forward runs in capacity mode (no host sync); the Adam step count and a sticky convergence flag live on the
      device, so a replay that runs after convergence changes nothing.
    * The frame being tracked is copied into a static viewpoint whose buffers the graph points at; the map is constant
      during tracking, so its activations are evaluated once and it takes no gradient.
    * The convergence flag is read back through a pinned buffer after every replay (or one replay late, `lookahead`).
    Result: identical poses and iteration counts to the eager loop with its per-iteration `if converged: break`."""

    def __init__(self, proto: Viewpoint, intr, gmap, bg, exclusive: bool = False):
        from . import rasterizer as _r
        self._r = _r
        dev = proto.device
        with torch.no_grad():
            self.map = (gmap.get_xyz.detach(), gmap.get_rotation.detach(), gmap.get_scaling.detach(),   # [P,1]: isotropic
                        gmap.get_opacity.detach(), gmap.get_features.detach())
        self.n_gaussians = int(self.map[0].shape[0])
        self.svp = Viewpoint(-1, torch.zeros_like(proto.rgb), torch.ones_like(proto.depth), dev)
        self.opt = PoseAdam(self.svp, 0.003, 0.001, 0.01, sticky=True)
        self.intr, self.bg = intr, bg
        self.flags = [torch.zeros(1, pin_memory=True) for _ in range(2)]
        self.events = [torch.cuda.Event() for _ in range(2)]
        self.graph = None
        self.zero2d = torch.zeros_like(self.map[0])
        # the static viewpoint's camera tensors: computed when a frame is loaded, then kept current by the pose step itself
        self.cam3 = (torch.empty(4, 4, device=dev), torch.empty(4, 4, device=dev), torch.empty(3, device=dev))
        self._load(proto)
        keep = (self.svp.R.clone(), self.svp.T.clone(), self.svp.exposure_a.data.clone(), self.svp.exposure_b.data.clone())
        # eager warm-up on a side stream (also records the capacity hint for this map size), then capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._iteration()
        torch.cuda.current_stream().wait_stream(s)
        self.opt.zero_grad()
        # two executable graphs of the same iteration, replayed alternately: launching a graph that is still running
        # waits for it, a second instance lets replay n+1 queue behind replay n (the ~30 us launch gap disappears).
        # `exclusive`: this process owns the device and the replays run one after the other on one stream, so the small sorts
        # may skip their ticket atomics (MGS_FLAG_EXCLUSIVE_DEVICE) -- NOT what a tracker beside a mapper process may assume.
        self.graphs = []
        for slot in range(2):
            self.opt.zero_grad()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), _r.exclusive_device(exclusive):
                self._iteration(host_flag=self.flags[slot])      # graph `slot` reports into its own pinned word
            self.graphs.append(g)
        self.graph = self.graphs[0]
        with torch.no_grad():          # undo the warm-up step
            self.svp.R.copy_(keep[0]); self.svp.T.copy_(keep[1])
            self.svp.exposure_a.data.copy_(keep[2]); self.svp.exposure_b.data.copy_(keep[3])

    def _iteration(self, host_flag=None):
        # render() without what tracking never reads: no screen-space gradient holder, no visibility filter
        xyz, rot, sca3, opa, col = self.map
        view, full, campos = self.cam3
        rs = GaussianRasterizationSettings(
            image_height=int(self.intr.height), image_width=int(self.intr.width),
            tanfovx=math.tan(self.intr.FoVx * 0.5), tanfovy=math.tan(self.intr.FoVy * 0.5), bg=self.bg, scale_modifier=1.0,
            viewmatrix=view, projmatrix=full, projmatrix_raw=self.intr.projection_matrix, sh_degree=0, campos=campos,
            prefiltered=False, debug=False)
        color, _, depth, opacity, _ = GaussianRasterizer(rs)(
            means3D=xyz, means2D=self.zero2d, opacities=opa, colors_precomp=col, scales=sca3, rotations=rot,
            theta=self.svp.cam_rot_delta, rho=self.svp.cam_trans_delta)
        self.opt.zero_grad()
        # loss value + upstream gradients in two launches, then the rasteriser's backward directly: no autograd node for
        # the scalar (its finalize kernel and the ones-fill of loss.backward() were two of the 25 launches of a replay)
        lg = fused_losses.loss_grads(color, depth, opacity, self.svp, tracking=True)
        lg.backward(color, depth, self.svp)
        self.opt.step_and_retract(sync=False, host_flag=host_flag, camera=(self.intr.projection_matrix,) + self.cam3)

    @torch.no_grad()
    def _load(self, vp: Viewpoint):
        s = self.svp
        s.rgb.copy_(vp.rgb); s.depth.copy_(vp.depth); s.mask.copy_(vp.mask); s.grad_mask.copy_(vp.grad_mask)
        s.R.copy_(vp.R); s.T.copy_(vp.T)
        s.exposure_a.data.copy_(vp.exposure_a.data); s.exposure_b.data.copy_(vp.exposure_b.data)
        s.cam_rot_delta.data.zero_(); s.cam_trans_delta.data.zero_()
        cam.fused_camera_matrices(s.R, s.T, self.intr.projection_matrix, out=self.cam3)
        self.opt.reset()
        for f in self.flags:          # (host words; nothing is in flight between two frames)
            f.zero_()

    def track(self, vp: Viewpoint, max_iters: int, lookahead: int = 1) -> int:
        """lookahead = 0: read the convergence flag after every replay (one 4-byte read-back per iteration).
        lookahead = 1: launch replay n before reading the flag of replay n-1 (hides the read-back; relies on the sticky
        flag making the surplus replay a no-op)."""
        self._load(vp)
        n_done = max_iters
        for n in range(max_iters):
            self.graphs[n & 1].replay()          # its pose step stores the convergence flag into self.flags[n & 1] (pinned)
            self.events[n & 1].record()
            m = n - lookahead
            if m >= 0:
                self.events[m & 1].synchronize()
                if float(self.flags[m & 1][0]) > 0.5:       # iteration m converged; any later replay was a no-op
                    n_done = m + 1
                    break
        torch.cuda.current_stream().synchronize()
        if self._r.check_overflow():
            raise RuntimeError("binning capacity overflow inside the captured tracking graph")
        with torch.no_grad():
            vp.R, vp.T = self.svp.R.clone(), self.svp.T.clone()
            vp.exposure_a.data.copy_(self.svp.exposure_a.data); vp.exposure_b.data.copy_(self.svp.exposure_b.data)
        return n_done

    def close(self):
        self._r.clear_graph_flags()
        self.graph = None
        self.graphs = []


def make_sequence(n_frames: int, intrinsics="fr3_office", n_gaussians=60000, seed=11, device="cuda:0"):
    """Ground-truth map + a smooth camera path; frames rendered by the rasteriser itself."""
    sc = make_scene(n_gaussians, intrinsics, seed=seed, near_fraction=0.0, mean_radius_px=9.0, device=device)
    intr = Intrinsics(sc.intr, device)
    gt = GaussianMap(device)
    gt._xyz, gt._rgb = sc.means3D, sc.colors
    gt._opacity = torch.logit(sc.opacities.clamp(0.05, 0.95) * 0 + 0.9)     # mostly opaque surface-like splats
    gt._scaling, gt._rotation = torch.log(sc.scales), sc.rotations
    bg = torch.zeros(3, device=device)
    T0 = torch.eye(4, device=device)
    T0[:3, :3], T0[:3, 3] = sc.R, sc.t
    frames: List[Viewpoint] = []
    with torch.no_grad():
        for i in range(n_frames):
            d = cam.se3_exp(torch.tensor([0.004 * i, -0.002 * i, 0.001 * i, 0.0, 0.0015 * i, 0.0005 * i], device=device))
            Tm = d @ T0
            vp = Viewpoint(i, torch.zeros(3, intr.height, intr.width, device=device),
                           torch.ones(intr.height, intr.width, device=device), device)
            vp.update_RT(Tm[:3, :3], Tm[:3, 3])
            pkg = _render(vp, intr, gt, bg)
            depth = torch.where(pkg["opacity"][0] > 0.5, pkg["depth"][0] / pkg["opacity"][0].clamp_min(1e-6),
                                torch.zeros_like(pkg["depth"][0]))
            frames.append(Viewpoint(i, pkg["render"].clamp(0, 1), depth, device, gt_R=Tm[:3, :3], gt_T=Tm[:3, 3]))
    return frames, intr


# ---- an OPAQUE-surface stand-in: a box room with furniture, ray-cast analytically -----------------------------------------
# The cloud of `make_sequence` is semi-transparent by construction (its "depth" is a blend over several layers), so a map
# fitted to it never gets past the reference's 0.7 opacity pruning threshold.  A real sequence shows opaque surfaces: this
# one is a 6 x 3 x 6 m room with four boxes standing in it, every pixel's colour and z-depth computed in closed form
# (ray / axis-aligned-box intersection, a procedural texture of the hit point), seen from a hand-held-like camera path.
_ROOM_HALF = (3.0, 1.5, 3.0)
_ROOM_BOXES = (   # (lo, hi) in world metres; y points down in the first camera, the floor is y = +1.5
    ((-2.4, 0.55, 1.1), (-0.9, 1.5, 2.3)),      # a desk
    ((0.9, -0.3, 1.8), (1.9, 1.5, 2.8)),        # a cabinet
    ((-0.5, 0.9, 0.9), (0.4, 1.5, 1.6)),        # a crate in front
    ((2.2, 0.2, -0.5), (3.0, 1.5, 0.9)),        # a shelf on the right wall
)


def _room_texture(p, axis, sid):
    """Colour of the surface point ``p`` [N,3] whose normal is along ``axis`` [N]; ``sid`` [N] picks the base colour."""
    dev = p.device
    ia = torch.where(axis == 0, 1, 0)
    ib = torch.where(axis == 2, 1, 2)
    a = torch.gather(p, 1, ia[:, None])[:, 0]
    b = torch.gather(p, 1, ib[:, None])[:, 0]
    base = torch.tensor([[0.78, 0.72, 0.62], [0.55, 0.66, 0.80], [0.70, 0.80, 0.62], [0.82, 0.60, 0.58], [0.60, 0.60, 0.72],
                         [0.85, 0.80, 0.55], [0.50, 0.72, 0.70], [0.75, 0.55, 0.75], [0.62, 0.78, 0.85], [0.80, 0.68, 0.50]],
                        device=dev)[sid % 10]
    two_pi = 2.0 * math.pi
    ph = sid.to(torch.float32) * 1.7
    slow = 0.5 + 0.5 * torch.sin(two_pi * 0.45 * a + ph) * torch.sin(two_pi * 0.38 * b + 0.6 * ph)
    # a soft checker (edges 3 cm wide) and a fine weave: image gradients everywhere, as a textured office has
    chk = torch.tanh(torch.sin(two_pi * a / 0.8) * torch.sin(two_pi * b / 0.8) * 12.0)
    fine = torch.sin(two_pi * a / 0.11 + ph) * torch.sin(two_pi * b / 0.13)
    lum = 0.62 + 0.16 * slow + 0.14 * chk + 0.06 * fine
    tint = torch.stack([torch.sin(two_pi * 0.21 * a + ph), torch.sin(two_pi * 0.17 * b + 2.0 + ph),
                        torch.sin(two_pi * 0.13 * (a + b) + 4.0)], 1) * 0.08
    return (base * lum[:, None] + tint).clamp(0.02, 0.98)


@torch.no_grad()
def raycast_room(R, t, k, device):
    """(rgb [3,H,W], depth [H,W]) of the room seen by the world->camera pose (R, t).  Pixel (x, y) looks along
    ((x + 0.5 - cx) / fx, (y + 0.5 - cy) / fy, 1): the rasteriser's pixel convention (``px = fx X/Z + cx - 0.5``) and the
    back-projection's (/root/reference/gaussian_splatting/scene/gaussian_model.py:232-236)."""
    H, W = k["H"], k["W"]
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    dc = torch.stack([(xs + 0.5 - k["cx"]) / k["fx"], (ys + 0.5 - k["cy"]) / k["fy"], torch.ones_like(xs)], -1).reshape(-1, 3)
    R, t = R.to(device), t.to(device)
    o = -(R.t() @ t)
    d = dc @ R                                            # R^T d, row form
    d = torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
    half = torch.tensor(_ROOM_HALF, device=device)
    t_wall = (torch.where(d > 0, half, -half) - o) / d    # the room from inside: the nearest exit plane
    best, axis = t_wall.min(dim=1)
    sid = axis * 2 + (torch.gather(d, 1, axis[:, None])[:, 0] > 0).long()
    for bi, (lo, hi) in enumerate(_ROOM_BOXES):
        lo, hi = torch.tensor(lo, device=device), torch.tensor(hi, device=device)
        t1, t2 = (lo - o) / d, (hi - o) / d
        tn, ax = torch.minimum(t1, t2).max(dim=1)
        tf = torch.maximum(t1, t2).min(dim=1).values
        hit = (tn < tf) & (tn > 1e-3) & (tn < best)
        best = torch.where(hit, tn, best)
        axis = torch.where(hit, ax, axis)
        sid = torch.where(hit, 6 + bi * 3 + ax, sid)
    p = o + best[:, None] * d
    rgb = _room_texture(p, axis, sid)
    return rgb.t().reshape(3, H, W).contiguous(), best.reshape(H, W).contiguous()


def make_room_sequence(n_frames: int, intrinsics="fr3_office", device="cuda:0", step_scale: float = 1.0):
    """``n_frames`` RGB-D frames of the room along a smooth hand-held-like path (about 1 cm and 0.3 degrees per frame at
    ``step_scale`` 1: the inter-frame motion of a 30 Hz TUM sequence), ground-truth poses attached."""
    k = dict(cam.INTRINSICS[intrinsics]) if isinstance(intrinsics, str) else dict(intrinsics)
    intr = Intrinsics(k, device)
    frames: List[Viewpoint] = []
    for i in range(n_frames):
        s = step_scale * i
        c = torch.tensor([0.35 * math.sin(0.022 * s) - 0.2, 0.05 * math.sin(0.05 * s) + 0.1, -1.6 + 0.25 * (1 - math.cos(0.02 * s))])
        yaw, pitch = 0.0055 * s - 0.1, 0.05 + 0.03 * math.sin(0.04 * s)
        Rwc = cam.so3_exp(torch.tensor([0.0, yaw, 0.0])) @ cam.so3_exp(torch.tensor([pitch, 0.0, 0.0]))   # camera -> world
        Rcw = Rwc.t().contiguous()
        tcw = -(Rcw @ c)
        rgb, depth = raycast_room(Rcw, tcw, k, device)
        frames.append(Viewpoint(i, rgb, depth, device, gt_R=Rcw.to(device), gt_T=tcw.to(device)))
    return frames, intr


def reference_style_tracking_loss(render_image, render_depth, render_opacity, viewpoint):
    """``get_loss_tracking`` in plain PyTorch ops, as the unmodified caller runs it (/root/reference/utils/slam_utils.py:58-98,
    ``invert_depth=False``): what an eager loop that swaps ONLY the rasteriser pays between the forward and the backward."""
    gt_depth = viewpoint.depth[None]
    opacity_mask = render_opacity > 0.99
    rgb = torch.exp(viewpoint.exposure_a) * render_image + viewpoint.exposure_b
    rgb_mask = viewpoint.mask * viewpoint.grad_mask * opacity_mask
    l1_rgb = (render_opacity * torch.abs(rgb * rgb_mask - viewpoint.rgb * rgb_mask).mean()).mean()
    depth_mask = (gt_depth > 0) * opacity_mask
    if depth_mask.any():
        l1_depth = torch.abs(render_depth[depth_mask] - gt_depth[depth_mask]).mean()
    else:
        l1_depth = torch.zeros((), device=render_depth.device)
    return 0.5 * l1_rgb + l1_depth


def eager_tracking_probe(frames, intr, gmap, bg, iters: int, profile_flavour=None):
    """The rate an UNMODIFIED MonoGS tracker gets from the drop-in: the loop of /root/reference/utils/slam_tracker.py:138-176
    -- ``render()`` through the seam (exact instance count: one read-back per forward, as upstream), the map's tensors
    requiring grad as the tracker's copy of the Gaussians does, ``loss.backward()``, ``torch.optim.Adam`` on the four pose /
    exposure parameters, ``update_pose`` -- with no hipGraph, no capacity mode.  Flavours, each a superset of the one before:
    ``torch_losses``      swaps ONLY the rasteriser: the loss is the reference's own torch ops (with their boolean-index
                          syncs), the pose step ``torch.optim.Adam`` + ``update_pose`` in torch ops (a host read-back each);
    ``fused_losses``      + ``monogs_amd.fused_losses.get_loss_tracking`` (same signature, two launches);
    ``fused_pose_step``   + ``PoseAdam.step_and_retract`` (Adam + retraction + camera tensors in one launch);
    ``render_loss_backward`` render + fused loss + backward, no pose step, with the device span of the same iterations;
    ``seam_only``         the same through the drop-in seam ALONE: the five map tensors handed over as already-activated
                          leaves, so that autograd stops at the rasteriser (no normalize / exp / sigmoid kernels and their
                          backward: those belong to the caller's GaussianModel getters) -- what tools/host_overhead.py times.
    Fixed iteration count (no early exit), pose and exposure restored afterwards."""
    import os
    from . import rasterizer as _r
    profile_flavour = profile_flavour or os.environ.get("MGS_PROBE_PROFILE")
    was = _r.sync_free_enabled()
    _r.set_sync_free(False)
    vp = frames[-1]
    keep = (vp.R.clone(), vp.T.clone(), vp.exposure_a.data.clone(), vp.exposure_b.data.clone())
    out = {}

    def map_tensors():
        return (gmap.get_xyz, gmap.get_rotation, gmap.get_scaling, gmap.get_opacity, gmap.get_features)

    def restore():
        with torch.no_grad():
            vp.update_RT(keep[0].clone(), keep[1].clone())
            vp.exposure_a.data.copy_(keep[2]); vp.exposure_b.data.copy_(keep[3])
            vp.cam_rot_delta.data.zero_(); vp.cam_trans_delta.data.zero_()
        for p in gmap.params():
            p.grad = None
    try:
        leaves = None
        for name in ("torch_losses", "fused_losses", "fused_pose_step", "render_loss_backward", "seam_only"):
            if name == "seam_only":
                with torch.no_grad():
                    leaves = [t.detach().clone().requires_grad_(True) for t in map_tensors()]
            loss_fn = reference_style_tracking_loss if name == "torch_losses" else fused_losses.get_loss_tracking
            if name in ("torch_losses", "fused_losses"):
                opt = torch.optim.Adam([dict(params=[vp.cam_rot_delta], lr=0.003), dict(params=[vp.cam_trans_delta], lr=0.001),
                                        dict(params=[vp.exposure_a], lr=0.01), dict(params=[vp.exposure_b], lr=0.01)])
                zero = opt.zero_grad
            else:
                popt = PoseAdam(vp, 0.003, 0.001, 0.01)
                zero = popt.zero_grad

            def it():
                zero()
                pkg = render(vp, intr, *(leaves if leaves is not None else map_tensors()), bg)
                loss = loss_fn(pkg["render"], pkg["depth"], pkg["opacity"], vp)
                loss.backward()
                if leaves is not None:
                    for t in leaves:
                        t.grad = None
                with torch.no_grad():
                    if name in ("torch_losses", "fused_losses"):
                        opt.step()
                        vp.retract()
                    elif name == "fused_pose_step":
                        popt.step_and_retract()
            # (un-timed iterations first, enough of them for the device to settle in the power state this loop keeps it in:
            #  behind a host-bound flavour it idles most of the time, and the first ~40 ms of load after that run slow)
            for _ in range(max(10, iters // 2)):
                it()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(iters):
                it()
            e1.record()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name] = dict(iters=iters, ms_per_iter=round(1e3 * dt / iters, 4), iters_per_s=round(iters / dt, 1))
            if profile_flavour == name:        # where the host time of this flavour goes (diagnostic)
                import cProfile
                import pstats
                import sys
                pr = cProfile.Profile()
                pr.enable()
                for _ in range(iters):
                    it()
                torch.cuda.synchronize()
                pr.disable()
                pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
            restore()
        # device time of render + loss + backward alone: the same iteration with the host queued ahead (capacity mode)
        _r.set_sync_free(True)
        popt = PoseAdam(vp, 0.003, 0.001, 0.01)

        def it_dev():
            popt.zero_grad()
            pkg = render(vp, intr, *leaves, bg)
            fused_losses.get_loss_tracking(pkg["render"], pkg["depth"], pkg["opacity"], vp).backward()
            for t in leaves:
                t.grad = None
        for _ in range(max(10, iters // 2)):
            it_dev()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            it_dev()
        e1.record()
        torch.cuda.synchronize()
        out["seam_only"]["device_ms_per_iter"] = round(e0.elapsed_time(e1) / iters, 4)
        _r.set_sync_free(False)
        with _r.collect_timing() as sink:          # one exact iteration with HIP events between the stages: what the device does
            it_dev()
            torch.cuda.synchronize()
        st = {}
        for d in sink:
            st.update({k: round(v, 4) for k, v in d.items() if k.endswith("_ms") and v > 0})
            if d.get("kind") == "forward":
                out["seam_only"]["num_rendered"] = int(d["num_rendered"])
        out["seam_only"]["stages_ms"] = st
        _r.check_overflow()
        restore()
    finally:
        _r.set_sync_free(was)
    out["gaussians"], out["width"], out["height"] = len(gmap), int(intr.width), int(intr.height)
    out["note"] = ("eager, exact instance count (one read-back per forward), map tensors require grad (ten-sum backward), fixed "
                   "iteration count against the final map of the run")
    return out


def run_slam(n_frames=12, intrinsics="fr3_office", tracking_itr_num=100, mapping_itr_num=150, window_size=8,
             kf_interval=4, init_itr_num=300, n_gaussians=60000, device="cuda:0", log=None,
             init_downsample=8, kf_downsample=16, point_size=1.0, graph_tracking=False, graph_mapping=False,
             track_lookahead=1, map_surgery=False, reference_lrs=False, prune_after_mapping=None,
             scene="cloud", reference_densify=False, eager_probe=0, exclusive_device=False):
    """Returns a dict with tracking / mapping FPS, iterations and the trajectory error.

    Mapping runs through ``monogs_amd.mapping.WindowMapper`` -- the SAME ``optimize_map`` / ``initialize_map`` the sharded
    window uses (render every window keyframe with screen-space gradient holder / radii / n_touched, fused losses, one
    backward, per-keyframe densification statistics + ``max_radii_2d`` + occlusion-aware visibility, fused Adam +
    learning-rate schedule, pose steps), replayed from hipGraphs when ``graph_mapping``.  Per keyframe, as ``Mapper.run``
    does (/root/reference/utils/slam_mapper.py:639-722): extend the map, fresh keyframe optimisers, ``optimize_map(iters)``,
    then ``optimize_map(prune=True, iters=1)`` (``prune_after_mapping``; default: with ``map_surgery``).
    ``map_surgery``: densify_and_prune / opacity resets / covisibility pruning on the reference's schedule
    (/root/reference/utils/slam_mapper.py:408-451,462-480).  ``reference_lrs``: the reference's learning rates and xyz schedule
    (``gaussian_map.REFERENCE_LRS``) instead of the harness's historical ones.
    ``scene``: "room" = opaque surfaces ray-cast analytically (``make_room_sequence``; survives the reference's 0.7 opacity
    pruning), "cloud" = the semi-transparent random cloud of ``make_sequence`` (historical; does not).
    ``reference_densify``: new Gaussians as the fork hard-codes them -- 1/32 of the pixels at initialisation, 1/64 per
    keyframe, scale^2 = dist2 x min(0.05, 0.01 x median depth) (/root/reference/gaussian_splatting/scene/gaussian_model.py:166-178)
    -- instead of ``init_downsample`` / ``kf_downsample`` / ``point_size``.
    ``eager_probe`` > 0: after the run, that many tracking iterations of the UNMODIFIED caller loop
    (/root/reference/utils/slam_tracker.py:138-176) against the final map, timed (``eager_tracking`` in the result)."""
    from .gaussian_map import REFERENCE_LRS, REFERENCE_LR_SCHEDULE
    from .mapping import WindowMapper
    if prune_after_mapping is None:
        prune_after_mapping = bool(map_surgery)
    if scene == "room":
        frames, intr = make_room_sequence(n_frames, intrinsics, device=device)
    else:
        frames, intr = make_sequence(n_frames, intrinsics, n_gaussians, device=device)
    if reference_densify:
        extend_kw = lambda init: dict(downsample=32 if init else 64, point_size=None)  # noqa: E731
    else:
        extend_kw = lambda init: dict(downsample=init_downsample if init else kf_downsample, point_size=point_size)  # noqa: E731
    bg = torch.zeros(3, device=device)
    gmap = GaussianMap(device, **(dict(lrs=REFERENCE_LRS) if reference_lrs else {}))
    if reference_lrs:
        gmap.lr_schedule = dict(REFERENCE_LR_SCHEDULE)
    gmap.surgery_log = []
    mapper = WindowMapper(gmap, intr, bg, window_size=window_size, use_graph=graph_mapping)
    mapper.map_surgery = bool(map_surgery)
    mapper.time_replays = True
    window: List[Viewpoint] = []
    per_frame, map_loss, window_sizes = [], [], []      # (frame, tracking iterations); (first, last) mapping loss per call
    size_trace = []                                     # (frame, Gaussians in the map after that keyframe's mapping)
    stats = dict(kf_extend_s=0.0, track_capture_s=0.0, track_s=0.0, track_iters=0, tracked=0, map_s=0.0, map_iters=0,
                 keyframes=0, renders=0)

    def sync():
        torch.cuda.synchronize()

    def map_window(iters, init=False):
        """One ``Mapper`` keyframe: the optimisation call, then the pruning call."""
        it0 = mapper.nr_iters
        if init:
            mapper.initialize_map(window[0], iters=1)
            first = mapper.last_loss
            if iters > 1:
                mapper.initialize_map(window[0], iters=iters - 1)
        else:
            mapper.new_keyframe_optimizers(window)
            mapper.optimize_map(window, iters=1)
            first = mapper.last_loss
            if iters > 1:
                mapper.optimize_map(window, iters=iters - 1)
        last = mapper.last_loss
        if not init and prune_after_mapping:
            mapper.optimize_map(window, prune=True, iters=1)
        if first is not None and last is not None:
            map_loss.append((first, last))
        stats["map_iters"] += mapper.nr_iters - it0
        stats["renders"] += (mapper.nr_iters - it0) * len(window)
        window_sizes.append(len(window))

    tgraph = None
    loss = torch.zeros(())
    for i, vp in enumerate(frames):
        if i == 0:
            vp.update_RT(vp.R_gt, vp.T_gt)
            sync(); t0 = time.perf_counter()
            gmap.extend_from_frame(vp, intr, init=True, **extend_kw(True))
            window.append(vp)
            map_window(init_itr_num, init=True)
            size_trace.append((0, len(gmap)))
            sync(); stats["map_s"] += time.perf_counter() - t0
            stats["keyframes"] += 1
            continue
        # ---- tracking (pose only; /root/reference/utils/slam_tracker.py:83-193)
        prev = frames[i - 1]
        vp.update_RT(prev.R.clone(), prev.T.clone())     # the fused pose step updates R, T in place
        sync(); t0 = time.perf_counter()
        if graph_tracking:
            if tgraph is None:                       # the map changed (or first frame): capture against the new map
                tgraph = TrackingGraph(vp, intr, gmap, bg, exclusive=exclusive_device)     # only a caller that owns the box may vouch for it
                sync(); stats["track_capture_s"] += time.perf_counter() - t0
            n_it = tgraph.track(vp, tracking_itr_num, lookahead=track_lookahead)
        else:
            opt = PoseAdam(vp, 0.003, 0.001, 0.01)
            n_it = 0
            for it in range(tracking_itr_num):
                pkg = _render(vp, intr, gmap, bg)
                opt.zero_grad()
                loss = fused_losses.get_loss_tracking(pkg["render"], pkg["depth"], pkg["opacity"], vp)
                loss.backward()
                n_it += 1
                with torch.no_grad():
                    if opt.step_and_retract():
                        break
        sync(); stats["track_s"] += time.perf_counter() - t0
        stats["track_iters"] += n_it
        stats["renders"] += n_it
        stats["tracked"] += 1
        per_frame.append((i, n_it))
        for p in gmap.params():
            p.grad = None
        # ---- keyframe + mapping
        if i % kf_interval == 0:
            sync(); t0 = time.perf_counter()
            with torch.no_grad():
                pkg = _render(vp, intr, gmap, bg)
            sync(); te0 = time.perf_counter()
            gmap.extend_from_frame(vp, intr, render_opacity=pkg["opacity"], render_depth=pkg["depth"], **extend_kw(False))
            sync(); stats["kf_extend_s"] += time.perf_counter() - te0
            window.append(vp)
            if len(window) > window_size:
                window.pop(1)
            map_window(mapping_itr_num)
            sync(); stats["map_s"] += time.perf_counter() - t0
            stats["keyframes"] += 1
            size_trace.append((i, len(gmap)))
            if tgraph is not None:                   # the map changed: the captured tracking graph is stale
                tgraph.close()
                tgraph = None
        if log:
            e = (-(vp.R.t() @ vp.T) + (vp.R_gt.t() @ vp.T_gt)).norm().item()
            with torch.no_grad():
                cov = (_render(vp, intr, gmap, bg)["opacity"] > 0.99).float().mean().item()
            log(f"frame {i}: P={gmap.get_xyz.shape[0]} track_iters={stats['track_iters']} kf={stats['keyframes']} "
                f"pos_err={e:.4f} m  opaque>0.99={cov:.2f} last_loss={float(loss):.5f}")

    if graph_tracking and tgraph is not None:
        tgraph.close()
    err = torch.stack([(-(f.R.t() @ f.T) + (f.R_gt.t() @ f.T_gt)).norm() for f in frames[1:]])
    ms = mapper.stats
    out = dict(stats)
    sl = gmap.surgery_log
    out["surgery"] = dict(
        densify_and_prune_calls=len(sl), cloned=sum(e["cloned"] for e in sl), split_net=sum(e["split_net"] for e in sl),
        pruned=sum(e["pruned"] for e in sl), calls_that_grew=sum(1 for e in sl if e["cloned"] + e["split_net"] > 0),
        calls_that_pruned=sum(1 for e in sl if e["pruned"] > 0), covisibility_prunes=len(mapper.coviz_log),
        covisibility_pruned=sum(n for _, n in mapper.coviz_log),
        gaussians_after_keyframe=[n for _, n in size_trace], log=sl[:6] + sl[-4:] if len(sl) > 10 else sl)
    if eager_probe:
        # (the captured graphs of the run and their private pools go first: the probe measures an eager caller, not one that
        #  shares its process with a few dozen instantiated hipGraphs)
        import gc
        mapper._drop_plan()
        mapper._pool = None
        gc.collect()
        torch.cuda.empty_cache()
        out["eager_tracking"] = eager_tracking_probe(frames, intr, gmap, bg, int(eager_probe))
    out.update(frames=n_frames, gaussians=int(gmap.get_xyz.shape[0]), width=intr.width, height=intr.height,
               tracking_fps=stats["tracked"] / max(stats["track_s"], 1e-9),
               tracking_iters_per_s=stats["track_iters"] / max(stats["track_s"], 1e-9),
               mapping_iters_per_s=stats["map_iters"] / max(stats["map_s"], 1e-9),
               mapping_kf_per_s=stats["keyframes"] / max(stats["map_s"], 1e-9),
               # steady state: graph replays only (no capture, no keyframe insertion, no one-time lazy loading)
               tracking_steady_iters_per_s=(stats["track_iters"] / max(stats["track_s"] - stats["track_capture_s"], 1e-9)
                                            if graph_tracking else None),
               mapping_steady_iters_per_s=(ms["replays"] / max(ms.get("replay_s", 0.0), 1e-9) if ms["replays"] else None),
               mapping_keyframe_iters_per_s=(ms.get("replay_kf", 0) / max(ms.get("replay_s", 0.0), 1e-9) if ms["replays"] else None),
               mapping_replays=ms["replays"], mapping_eager_iters=ms["eager_iters"], mapping_captures=ms["captures"],
               mapping_capture_s=ms["capture_s"], window_sizes=window_sizes,
               kf_extend_ms=1e3 * stats["kf_extend_s"] / max(stats["keyframes"] - 1, 1),
               ate_rmse_m=float(torch.sqrt((err ** 2).mean())),
               track_iters_per_frame=per_frame,
               poses=[(f.R.detach().cpu().clone(), f.T.detach().cpu().clone()) for f in frames],
               position_error_m=[float(e) for e in err],
               camera_centers=[(-(f.R.t() @ f.T)).cpu() for f in frames],
               camera_centers_gt=[(-(f.R_gt.t() @ f.T_gt)).cpu() for f in frames],
               map_loss=[(float(a), float(b)) for a, b in map_loss],
               graph_tracking=bool(graph_tracking), graph_mapping=bool(graph_mapping), map_surgery=bool(map_surgery),
               config=dict(tracking_itr_num=tracking_itr_num, mapping_itr_num=mapping_itr_num,
                           window_size=window_size, kf_interval=kf_interval, init_itr_num=init_itr_num))
    return out


# ---- the two-process topology of the reference: tracker in the main process, mapper in a spawned one ---------------------
# (/root/reference/slam.py:102-179: `mp.Process(target=self.mapper.run)`, queues between them; the map crosses the process
#  boundary on every keyframe -- there as `clone_obj(self.gaussians)` pickled through an mp.Queue,
#  /root/reference/utils/slam_mapper.py:550-564, here through `MapArena`: two pre-allocated device buffers per tensor shared
#  once over HIP IPC, a publish = device-to-device copies + a header store, an acquire = views.)
ARENA_FIELDS = {"xyz": (3,), "rotation": (4,), "scaling": (1,), "opacity": (1,), "rgb": (3,)}


class _ArenaMapView:
    """What the tracker needs of a map, over the views a `MapArena.acquire()` hands out (already activated)."""

    def __init__(self, views):
        self.get_xyz, self.get_rotation, self.get_scaling = views["xyz"], views["rotation"], views["scaling"]
        self.get_opacity, self.get_features = views["opacity"], views["rgb"]


def _mapper_process(arena, q_in, q_out, cfg):
    """`Mapper.run` in miniature (/root/reference/utils/slam_mapper.py:566-734): wait for `init` / `keyframe` / `stop`,
    extend the map from the keyframe, optimise the window, publish the map."""
    import multiprocessing
    from .mapping import WindowMapper
    dev = cfg["device"]
    torch.cuda.set_device(torch.device(dev))
    frames, intr = make_sequence(cfg["n_frames"], cfg["intrinsics"], cfg["n_gaussians"], device=dev)
    bg = torch.zeros(3, device=dev)
    gmap = GaussianMap(dev)
    mapper = WindowMapper(gmap, intr, bg, window_size=cfg["window_size"], use_graph=cfg["graph"])
    mapper.map_surgery = False
    window: List[Viewpoint] = []

    def publish():
        with torch.no_grad():
            return arena.publish({"xyz": gmap.get_xyz, "rotation": gmap.get_rotation, "scaling": gmap.get_scaling,
                                  "opacity": gmap.get_opacity, "rgb": gmap.get_features})
    try:
        while True:
            msg = q_in.get()
            if msg[0] == "stop":
                break
            tag, idx, R, T = msg
            vp = frames[idx]
            vp.update_RT(torch.tensor(R, device=dev), torch.tensor(T, device=dev))
            t0 = time.perf_counter()
            if tag == "init":
                gmap.extend_from_frame(vp, intr, downsample=cfg["init_downsample"], init=True, point_size=1.0)
                window.append(vp)
                mapper.initialize_map(vp, iters=cfg["init_itr_num"])
            else:
                with torch.no_grad():
                    pkg = _render(vp, intr, gmap, bg)
                gmap.extend_from_frame(vp, intr, downsample=cfg["kf_downsample"], render_opacity=pkg["opacity"], point_size=1.0)
                window.append(vp)
                if len(window) > cfg["window_size"]:
                    window.pop(1)
                mapper.new_keyframe_optimizers(window)
                mapper.optimize_map(window, iters=cfg["mapping_itr_num"])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            seq = publish()
            t2 = time.perf_counter()
            # (the keyframe's refined pose goes back with the answer, as `sync_backend` carries the keyframes back)
            q_out.put(("done", idx, seq, len(gmap), t1 - t0, t2 - t1, vp.R.cpu().numpy(), vp.T.cpu().numpy(), len(window)))
    finally:
        multiprocessing.current_process()._args = ()       # (a spawned child leaves through os._exit: drop the IPC mappings now)
        del arena
        import gc
        gc.collect()
        torch.cuda.ipc_collect()


def run_slam_two_process(n_frames=12, intrinsics="fr3_office", tracking_itr_num=100, mapping_itr_num=150, window_size=8,
                         kf_interval=4, init_itr_num=150, n_gaussians=60000, device="cuda:0", capacity=400000, graph=True,
                         init_downsample=8, kf_downsample=16):
    """The tracking / mapping loops of `run_slam` split over two processes as the reference runs them, with the map handed
    over through `MapArena`.  The tracker (this process) tracks every frame against the
    snapshot it last acquired and, on a keyframe, asks the mapper and waits for the next publish -- the reference's
    tracker does the same (`utils/slam_tracker.py:362-365`).  Returns rates, the hand-off times and the trajectory error."""
    import torch.multiprocessing as mp

    from .map_arena import MapArena
    ctx = mp.get_context("spawn")
    arena = MapArena(capacity, ARENA_FIELDS, device=device)
    q_in, q_out = ctx.Queue(), ctx.Queue()
    cfg = dict(n_frames=n_frames, intrinsics=intrinsics, n_gaussians=n_gaussians, device=device, window_size=window_size,
               init_itr_num=init_itr_num, mapping_itr_num=mapping_itr_num, graph=graph, init_downsample=init_downsample,
               kf_downsample=kf_downsample)
    proc = ctx.Process(target=_mapper_process, args=(arena, q_in, q_out, cfg))
    proc.start()
    frames, intr = make_sequence(n_frames, intrinsics, n_gaussians, device=device)
    bg = torch.zeros(3, device=device)
    stats = dict(track_s=0.0, track_iters=0, tracked=0, wait_s=0.0, acquire_s=0.0, publish_s=0.0, map_s=0.0, keyframes=0)
    windows, sizes, seqs = [], [], []
    tgraph, snapshot, seq = None, None, 0
    t_all = time.perf_counter()

    def ask(tag, vp):
        nonlocal snapshot, seq, tgraph
        t0 = time.perf_counter()
        q_in.put((tag, vp.frame_idx, vp.R.cpu().numpy(), vp.T.cpu().numpy()))
        import queue as _queue
        deadline = time.perf_counter() + 600.0
        while True:                       # a dead mapper must not leave the tracker blocked for ten minutes
            try:
                ans = q_out.get(timeout=1.0)
                break
            except _queue.Empty:
                if not proc.is_alive():
                    raise RuntimeError(f"the mapper process died (exit code {proc.exitcode}) while the tracker waited for keyframe "
                                       f"{vp.frame_idx}") from None
                if time.perf_counter() > deadline:
                    raise RuntimeError("the mapper process did not answer within 600 s") from None
        stats["wait_s"] += time.perf_counter() - t0
        _, idx, new_seq, P, t_map, t_pub, R, T, wlen = ans
        stats["map_s"] += t_map
        stats["publish_s"] += t_pub
        stats["keyframes"] += 1
        t1 = time.perf_counter()
        got, views = arena.acquire()
        stats["acquire_s"] += time.perf_counter() - t1
        assert got == new_seq and int(views["xyz"].shape[0]) == P, (got, new_seq, P)
        vp.update_RT(torch.tensor(R, device=device), torch.tensor(T, device=device))
        if tgraph is not None:
            tgraph.close()
            tgraph = None
        snapshot, seq = _ArenaMapView(views), got
        windows.append(wlen); sizes.append(P); seqs.append(got)
    try:
        for i, vp in enumerate(frames):
            if i == 0:
                vp.update_RT(vp.R_gt, vp.T_gt)
                ask("init", vp)
                continue
            prev = frames[i - 1]
            vp.update_RT(prev.R.clone(), prev.T.clone())
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if tgraph is None:
                tgraph = TrackingGraph(vp, intr, snapshot, bg, exclusive=False)      # the mapper process shares the device
            n_it = tgraph.track(vp, tracking_itr_num)
            torch.cuda.synchronize(); stats["track_s"] += time.perf_counter() - t0
            assert not arena.stale(seq), "the mapper overwrote the snapshot the tracker was reading"
            stats["track_iters"] += n_it
            stats["tracked"] += 1
            if i % kf_interval == 0:
                ask("keyframe", vp)
    finally:
        if tgraph is not None:
            tgraph.close()
        q_in.put(("stop",))
        proc.join(timeout=120)
        if proc.is_alive():               # hung: do not leave a child holding a HIP context and the IPC mappings behind
            proc.terminate()
            proc.join(timeout=10)
            if proc.is_alive():
                proc.kill()
                proc.join(timeout=10)
        snapshot = None
        del arena
        import gc
        for _ in range(5):
            gc.collect()
            torch.cuda.ipc_collect()
            time.sleep(0.05)
    wall = time.perf_counter() - t_all
    exitcode = proc.exitcode
    if exitcode != 0:
        raise RuntimeError(f"the mapper process ended with exit code {exitcode}")
    err = torch.stack([(-(f.R.t() @ f.T) + (f.R_gt.t() @ f.T_gt)).norm() for f in frames[1:]])
    k = max(stats["keyframes"], 1)
    return dict(stats, frames=n_frames, wall_s=wall, fps_end_to_end=(n_frames - 1) / wall,
                tracking_iters_per_s=stats["track_iters"] / max(stats["track_s"], 1e-9),
                tracking_fps=stats["tracked"] / max(stats["track_s"], 1e-9),
                handoff_ms=dict(publish=1e3 * stats["publish_s"] / k, acquire=1e3 * stats["acquire_s"] / k),
                mapper_busy_ms_per_keyframe=1e3 * stats["map_s"] / k, window_sizes=windows, gaussians=sizes, sequences=seqs,
                ate_rmse_m=float(torch.sqrt((err ** 2).mean())), exitcode=exitcode)
