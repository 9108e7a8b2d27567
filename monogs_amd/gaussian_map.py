"""The Gaussian map as the mapper sees it: parameters, activations, optimiser and the map surgery
(grow / clone / split / prune / opacity reset) that follows the rasteriser backward.

Host-side mirror of the subset of ``GaussianModel`` that ``Mapper.optimize_map`` drives
(/root/reference/gaussian_splatting/scene/gaussian_model.py:84-106 activations, :398-442 optimiser groups,
:522-535 opacity resets, :642-776 optimiser-state surgery, :778-892 densify / prune / statistics) for this
fork's isotropic RGB map (``_features_dc`` is ``[P,3]``, ``_scaling`` ``[P,1]``).  Same method names and argument
meaning; what changes is where the work runs:

* the optimiser is ``GaussianAdam`` (one fused launch, step counts on the device) and its state surgery is
  ``extend`` / ``prune`` / ``replace`` -- moments carried exactly as ``cat_tensors_to_optimizer`` /
  ``_prune_optimizer`` / ``replace_tensor_to_optimizer`` do;
* ``kf_idx`` / ``nr_obs`` stay on the device (the reference keeps them on the CPU and moves masks back and forth);
* ``densify_and_split`` draws its samples from a caller-supplied ``torch.Generator`` so that the replicas of a
  keyframe-sharded mapping window split identically on every rank (``window.split_generator``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .gaussian_optim import GaussianAdam, add_densification_stats, expon_lr


def inverse_sigmoid(x: torch.Tensor) -> torch.Tensor:
    return torch.log(x / (1 - x))


def build_rotation(q: torch.Tensor) -> torch.Tensor:
    """Rotation matrices of (r, x, y, z) quaternions, normalised first
    (/root/reference/gaussian_splatting/utils/general_utils.py:113-136)."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


# learning rates of the reference's five groups with spatial_lr_scale = 6.0 as the harness has always used them
# (/root/reference/configs/mono/tum/base_config.yaml:57-65 through gaussian_model.py:405-436)
DEFAULT_LRS = (1.6e-4 * 6.0, 0.0025, 0.05, 0.001, 0.001)
NAMES = ("xyz", "f_dc", "opacity", "scaling", "rotation")
# what the reference itself trains with: training_setup (gaussian_model.py:398-449) on opt_params of
# /root/reference/configs/mono/tum/base_config.yaml:49-66 with init_lr(6.0) (/root/reference/slam.py:87): position_lr_init
# 0.0016 and scaling_lr 0.001 are multiplied by spatial_lr_scale = 6; the xyz rate then follows the log-linear schedule
# of general_utils.helper (no delay: lr_delay_steps = 0) from 0.0096 to 0.0000096 over 30 000 mapping iterations
REFERENCE_LRS = (0.0016 * 6.0, 0.0025, 0.05, 0.001 * 6.0, 0.001)
REFERENCE_LR_SCHEDULE = dict(lr_init=0.0016 * 6.0, lr_final=0.0000016 * 6.0, lr_delay_steps=0, lr_delay_mult=0.01,
                             max_steps=30000)


class GaussianMap:
    """Isotropic RGB map with the reference's activations."""

    def __init__(self, device, capturable=False, fused_adam=True, lrs: Sequence[float] = DEFAULT_LRS,
                 percent_dense: float = 0.01):
        self.device = device
        self.capturable = capturable        # torch.optim.Adam(capturable=True): step counters on the device (hipGraph)
        self.fused_adam = fused_adam        # monogs_amd.gaussian_optim.GaussianAdam (one launch, always capturable)
        self.lrs = [float(x) for x in lrs]
        self.percent_dense = float(percent_dense)
        e = lambda *s: torch.empty(*s, device=device)  # noqa: E731
        self._xyz, self._rgb, self._opacity, self._scaling, self._rotation = e(0, 3), e(0, 3), e(0, 1), e(0, 1), e(0, 4)
        self.optimizer = None
        self._init_stats(0)
        self.kf_idx = torch.empty(0, dtype=torch.int32, device=device)
        self.nr_obs = torch.empty(0, dtype=torch.int32, device=device)
        # position learning-rate schedule (update_learning_rate, gaussian_model.py:451-465); off unless configured
        self.lr_schedule = None             # dict(lr_init, lr_final, lr_delay_mult, max_steps)
        self.surgery_log = None             # a list: densify_and_prune appends the map sizes around each of its steps

    get_xyz = property(lambda s: s._xyz)
    get_features = property(lambda s: s._rgb)
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))

    def __len__(self):
        return int(self._xyz.shape[0])

    def params(self) -> List[torch.Tensor]:
        return [self._xyz, self._rgb, self._opacity, self._scaling, self._rotation]

    def _set_params(self, ps):
        self._xyz, self._rgb, self._opacity, self._scaling, self._rotation = ps

    def _init_stats(self, P):
        z = lambda *s: torch.zeros(*s, device=self.device)  # noqa: E731
        self.xyz_gradient_accum, self.denom, self.max_radii_2d = z(P, 1), z(P, 1), z(P)

    # ---- growth: densification_postfix (gaussian_model.py:745-776) ---------------------------------------------
    @torch.no_grad()
    def densification_postfix(self, new_xyz, new_rgb, new_opacity, new_scaling, new_rotation, new_kf_idxs=None,
                              new_nr_obs=None):
        new = [new_xyz, new_rgb, new_opacity, new_scaling, new_rotation]
        n_new = int(new_xyz.shape[0])
        if self.optimizer is None:
            self._set_params([torch.cat([o.detach(), n.detach()], 0).requires_grad_(True)
                              for o, n in zip(self.params(), new)])
            self.optimizer = self._make_optimizer()
        elif self.fused_adam:
            self._set_params(self.optimizer.extend(new))
        else:
            self._torch_extend(new)
        self._init_stats(len(self))          # the reference zeroes all three statistics whenever the map grows
        i32 = dict(dtype=torch.int32, device=self.device)
        self.kf_idx = torch.cat((self.kf_idx, (new_kf_idxs if new_kf_idxs is not None
                                               else torch.zeros(n_new, **i32)).to(**i32)))
        self.nr_obs = torch.cat((self.nr_obs, (new_nr_obs if new_nr_obs is not None
                                               else torch.zeros(n_new, **i32)).to(**i32)))

    def _make_optimizer(self):
        if self.fused_adam:
            return GaussianAdam(self.params(), self.lrs, eps=1e-15)
        groups = [{"params": [p], "lr": lr, "name": n} for p, lr, n in zip(self.params(), self.lrs, NAMES)]
        try:       # one multi-tensor kernel per step (the reference uses the default, unfused Adam)
            return torch.optim.Adam(groups, eps=1e-15, fused=True, capturable=self.capturable)
        except (RuntimeError, TypeError, ValueError):      # this torch build has no fused Adam for the device / dtype
            return torch.optim.Adam(groups, eps=1e-15, capturable=self.capturable)

    def _torch_extend(self, new):
        old_state = self.optimizer.state_dict()["state"]
        self._set_params([torch.cat([o.detach(), n.detach()], 0).requires_grad_(True)
                          for o, n in zip(self.params(), new)])
        self.optimizer = self._make_optimizer()
        for i, p in enumerate(self.params()):
            st = old_state.get(i)
            if st is not None:
                k = p.shape[0] - st["exp_avg"].shape[0]
                pad = lambda t: torch.cat([t, torch.zeros(k, *t.shape[1:], device=self.device)], 0)  # noqa: E731
                self.optimizer.state[p] = dict(step=st["step"], exp_avg=pad(st["exp_avg"]),
                                               exp_avg_sq=pad(st["exp_avg_sq"]))

    # ---- prune_points (gaussian_model.py:682-707) -----------------------------------------------------------------
    @torch.no_grad()
    def prune_points(self, mask: torch.Tensor):
        assert self.fused_adam, "map surgery is implemented for the fused optimiser"
        keep = ~mask.to(self.device).bool()
        self._set_params(self.optimizer.prune(keep))
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.denom = self.denom[keep]
        self.max_radii_2d = self.max_radii_2d[keep]
        self.kf_idx = self.kf_idx[keep]
        self.nr_obs = self.nr_obs[keep]

    # ---- densify (gaussian_model.py:778-886) -------------------------------------------------------------------
    @torch.no_grad()
    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = torch.norm(grads, dim=-1) >= grad_threshold
        sel &= self.get_scaling.max(dim=1).values <= self.percent_dense * scene_extent
        self.densification_postfix(self._xyz[sel], self._rgb[sel], self._opacity[sel], self._scaling[sel],
                                   self._rotation[sel], new_kf_idxs=self.kf_idx[sel], new_nr_obs=self.nr_obs[sel])

    @torch.no_grad()
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, generator: Optional[torch.Generator] = None):
        n_init = len(self)
        padded = torch.zeros(n_init, device=self.device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = padded >= grad_threshold
        sel &= self.get_scaling.max(dim=1).values > self.percent_dense * scene_extent
        stds = self.get_scaling[sel].repeat(N, 1).expand(-1, 3)
        samples = torch.randn(stds.shape, device=self.device, generator=generator) * stds      # normal(0, stds)
        rots = build_rotation(self._rotation[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self._xyz[sel].repeat(N, 1)
        new_scaling = torch.log(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz, self._rgb[sel].repeat(N, 1), self._opacity[sel].repeat(N, 1), new_scaling,
                                   self._rotation[sel].repeat(N, 1), new_kf_idxs=self.kf_idx[sel].repeat(N),
                                   new_nr_obs=self.nr_obs[sel].repeat(N))
        prune = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=self.device, dtype=torch.bool)))
        self.prune_points(prune)

    @torch.no_grad()
    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        n0 = len(self)
        self.densify_and_clone(grads, max_grad, extent)
        n1 = len(self)
        self.densify_and_split(grads, max_grad, extent, generator=generator)
        n2 = len(self)
        prune = (self.get_opacity < min_opacity).squeeze(1)
        if max_screen_size:
            prune = prune | (self.max_radii_2d > max_screen_size) | (self.get_scaling.max(dim=1).values > 0.1 * extent)
        self.prune_points(prune)
        if self.surgery_log is not None:     # (map sizes only: no extra device work, the lengths are host integers)
            self.surgery_log.append(dict(before=n0, cloned=n1 - n0, split_net=n2 - n1, pruned=n2 - len(self), after=len(self)))

    def add_densification_stats(self, viewspace_point_tensor, radii):
        """One rendered keyframe: ``xyz_gradient_accum`` / ``denom`` (gaussian_model.py:888-892) and ``max_radii_2d``
        (/root/reference/utils/slam_mapper.py:453-457) over the visible Gaussians, one launch."""
        add_densification_stats(viewspace_point_tensor.grad, radii, self.xyz_gradient_accum, self.denom,
                                self.max_radii_2d)

    # ---- opacity resets (gaussian_model.py:522-535) ---------------------------------------------------------
    @torch.no_grad()
    def reset_opacity(self):
        new = inverse_sigmoid(torch.ones_like(self._opacity) * 0.01)
        self._opacity = self.optimizer.replace(2, new)

    @torch.no_grad()
    def reset_opacity_nonvisible(self, visibility_filters, as_reference: bool = True):
        """gaussian_model.py:527-535.  The reference writes the ACTIVATED opacity of the visible Gaussians into the raw
        parameter (``opacities_new[filter] = self.get_opacity[filter]``), so a visible Gaussian comes out of the reset with
        opacity sigmoid(sigmoid(x)) in [0.5, 0.73].  ``as_reference=True`` (default) reproduces that bit for bit -- a drop-in
        must give the reference's map; ``as_reference=False`` keeps the visible Gaussians' raw opacity (what the upstream
        MonoGS code base intends).  Pinned either way by tests/test_host_api.py."""
        new = inverse_sigmoid(torch.ones_like(self._opacity) * 0.4)
        src = self.get_opacity.detach() if as_reference else self._opacity.detach()
        for f in visibility_filters:
            new[f] = src[f]
        self._opacity = self.optimizer.replace(2, new)

    def update_learning_rate(self, iteration: int):
        if self.lr_schedule is None:
            return None
        lr = expon_lr(iteration, **self.lr_schedule)
        if self.fused_adam:
            self.optimizer.set_lr(0, lr)
        else:
            self.optimizer.param_groups[0]["lr"] = lr
        return lr

    # ---- new Gaussians from a keyframe (extend_from_pcd_seq, gaussian_model.py:321-396) --------------------------
    def extend_from_frame(self, vp, intr, downsample: int, point_size=0.05, init=False, render_opacity=None,
                          render_depth=None, kf_id: Optional[int] = None):
        """Back-project a keyframe's depth into new Gaussians; scale from distCUDA2
        (gaussian_model.py:121-319 via ``monogs_amd.keyframe``).  ``point_size=None``: the reference's rule,
        scale^2 = dist2 x min(0.05, 0.01 x median depth) (gaussian_model.py:173-178); a number: scale^2 = dist2 x point_size."""
        from .keyframe import create_viewpoint_pcd
        g = torch.Generator(device=self.device).manual_seed(1000 + vp.frame_idx)
        ps = dict(point_size=0.01, point_size_max=0.05) if point_size is None else dict(point_size=1e9, point_size_max=point_size)
        pw, rgb, scales, rots, opac, _ = create_viewpoint_pcd(
            vp, intr, render_depth=None if init else (render_depth if render_depth is not None else vp.depth),
            render_opacity=None if init else render_opacity, init=init,
            generator=g, downsample_factor=downsample, **ps)
        n_new = pw.shape[0]
        if n_new < 4:
            return 0
        kf = torch.full((n_new,), int(vp.frame_idx if kf_id is None else kf_id), dtype=torch.int32, device=self.device)
        if self.fused_adam or self.optimizer is None:
            # (extend_from_pcd_seq goes through densification_postfix as well: statistics restart from zero)
            self.densification_postfix(pw, rgb, opac, scales, rots, new_kf_idxs=kf)
        else:
            self._torch_extend([pw, rgb, opac, scales, rots])
            self._init_stats(len(self))
            self.kf_idx = torch.cat((self.kf_idx, kf))
            self.nr_obs = torch.cat((self.nr_obs, torch.zeros_like(kf)))
        return n_new
