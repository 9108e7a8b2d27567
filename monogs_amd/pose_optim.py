"""Fused pose optimiser step: Adam on the per-camera deltas / exposure + SE(3) retraction in one launch.

Equivalent to the reference's per-iteration sequence
    pose_optimizer.step(); converged = update_pose(viewpoint)
(/root/reference/utils/slam_tracker.py:172-174, utils/slam_mapper.py:486-496, utils/pose_utils.py:76-93)
with torch.optim.Adam defaults; checked against exactly that sequence in tests/test_gpu_pose.py.
"""
from __future__ import annotations

import torch

from . import _lib
from .rasterizer import _stream, _device_guard


class PoseAdam:
    def __init__(self, viewpoint, lr_rot=0.003, lr_trans=0.001, lr_exposure=0.01, betas=(0.9, 0.999), eps=1e-8,
                 sticky=False):
        self.vp = viewpoint
        self.sticky = bool(sticky)   # once converged, further steps are no-ops (graph-replayed tracking loops)
        self.lrs = (float(lr_rot), float(lr_trans), float(lr_exposure))
        self.betas, self.eps = betas, eps
        dev = viewpoint.cam_rot_delta.device
        self.m = torch.zeros(8, device=dev)
        self.v = torch.zeros(8, device=dev)
        self.out = torch.zeros(2, device=dev)
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=dev)     # Adam step count lives on the device

    @torch.no_grad()
    def reset(self):
        """Fresh optimiser state (what constructing a new torch.optim.Adam per frame does in the reference,
        /root/reference/utils/slam_tracker.py:113-140) without re-allocating: captured graphs keep pointing at it."""
        self.m.zero_(); self.v.zero_(); self.out.zero_(); self.t_dev.zero_()

    def zero_grad(self):
        vp = self.vp
        for p in (vp.cam_rot_delta, vp.cam_trans_delta, vp.exposure_a, vp.exposure_b):
            p.grad = None

    @torch.no_grad()
    def step_and_retract(self, converged_threshold=1e-4, sync=True, host_flag=None, camera=None):
        """Returns the convergence flag (bool) when ``sync`` else the device tensor out[2].  ``host_flag``: a PINNED
        host float tensor whose first element the kernel also writes the flag to (device stores into mapped host memory:
        a loop replayed from a hipGraph then needs no device-to-host copy per iteration).  ``camera``: the viewpoint's
        ``(projmatrix_raw, viewmatrix, projmatrix, campos)`` tensors (``camera.fused_camera_matrices``): the kernel
        refreshes the last three from the updated pose, bit-identically to ``mgs_camera_setup``, so the next render of
        the loop launches no camera kernel."""
        lib = _lib.load()
        vp = self.vp
        R = vp.R.contiguous() if not vp.R.is_contiguous() else vp.R
        T = vp.T.contiguous() if not vp.T.is_contiguous() else vp.T
        if R.data_ptr() != vp.R.data_ptr() or T.data_ptr() != vp.T.data_ptr():
            vp.R, vp.T = R, T
        if camera is None:        # camera tensors cached on the viewpoint by render(): keep them current in the same launch
            c = getattr(vp, "_mgs_cam", None)
            if c is not None and c[0] is vp.R and c[1] is vp.T:
                camera = (c[2],) + tuple(c[3])
        g = lambda p: None if p.grad is None else p.grad.contiguous().data_ptr()  # noqa: E731
        with _device_guard(R.device):
            _lib.check(lib.mgs_pose_step(R.data_ptr(), T.data_ptr(), vp.cam_rot_delta.data_ptr(),
                                         vp.cam_trans_delta.data_ptr(), vp.exposure_a.data_ptr(),
                                         vp.exposure_b.data_ptr(), g(vp.cam_rot_delta), g(vp.cam_trans_delta),
                                         g(vp.exposure_a), g(vp.exposure_b), self.m.data_ptr(), self.v.data_ptr(),
                                         0, self.lrs[0], self.lrs[1], self.lrs[2], self.betas[0], self.betas[1],
                                         self.eps, float(converged_threshold), self.t_dev.data_ptr(),
                                         self.out.data_ptr(), 1 if self.sticky else 0,
                                         None if host_flag is None else host_flag.data_ptr(),
                                         *((None,) * 4 if camera is None else tuple(t.data_ptr() for t in camera)),
                                         _stream()),
                       "mgs_pose_step")
        return bool(self.out[0].item() > 0.5) if sync else self.out

    def _pointers(self):
        """The 18 device pointers of this optimiser's update, in ``mgs_pose_step_batch`` order."""
        vp = self.vp
        if not vp.R.is_contiguous() or not vp.T.is_contiguous():
            vp.R, vp.T = vp.R.contiguous(), vp.T.contiguous()
        cam = (None,) * 4
        c = getattr(vp, "_mgs_cam", None)
        if c is not None and c[0] is vp.R and c[1] is vp.T:
            cam = (c[2].data_ptr(),) + tuple(t.data_ptr() for t in c[3])
        g = lambda p: None if p.grad is None else p.grad.contiguous().data_ptr()  # noqa: E731
        return (vp.R.data_ptr(), vp.T.data_ptr(), vp.cam_rot_delta.data_ptr(), vp.cam_trans_delta.data_ptr(),
                vp.exposure_a.data_ptr(), vp.exposure_b.data_ptr(), g(vp.cam_rot_delta), g(vp.cam_trans_delta),
                g(vp.exposure_a), g(vp.exposure_b), self.m.data_ptr(), self.v.data_ptr(), self.t_dev.data_ptr(),
                self.out.data_ptr()) + cam

    @staticmethod
    @torch.no_grad()
    def step_batch(optimisers, converged_threshold=1e-4):
        """``step_and_retract(sync=False)`` of several viewpoints in ONE launch (the keyframes of a mapping window; they must
        share learning rates, betas and eps, as the reference's parameter groups do, utils/slam_mapper.py:687-717).  Groups of
        16."""
        import ctypes as C
        optimisers = list(optimisers)
        if not optimisers:
            return
        lib = _lib.load()
        o0 = optimisers[0]
        for o in optimisers[1:]:
            if (o.lrs, o.betas, o.eps, o.sticky) != (o0.lrs, o0.betas, o0.eps, o0.sticky):
                raise ValueError("step_batch: the optimisers must share learning rates, betas, eps and the sticky flag")
        with _device_guard(o0.m.device):
            for i in range(0, len(optimisers), 16):
                grp = optimisers[i:i + 16]
                flat = [x for o in grp for x in o._pointers()]
                arr = (C.c_void_p * len(flat))(*flat)
                _lib.check(lib.mgs_pose_step_batch(len(grp), arr, o0.lrs[0], o0.lrs[1], o0.lrs[2], o0.betas[0], o0.betas[1],
                                                   o0.eps, float(converged_threshold), 1 if o0.sticky else 0, _stream()),
                           "mgs_pose_step_batch")
