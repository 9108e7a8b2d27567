"""``GaussianRasterizationSettings`` / ``GaussianRasterizer`` -- host side of the drop-in.

Mirrors the Python surface of the un-vendored ``diff_gaussian_rasterization`` (w-pose variant)
exactly as MonoGS uses it: settings built at
/root/reference/gaussian_splatting/gaussian_renderer/__init__.py:70-84, rasteriser called by
keyword at :130-156, outputs consumed at :160-168.  The arithmetic lives in
libmonogs_raster.so (HIP, gfx950) behind the C ABI of include/monogs_raster.h; torch is used
for device memory, the current stream and autograd plumbing only.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---- sync-free ("capacity") mode ----------------------------------------------------------------
# The exact path (the default: what an unmodified MonoGS gets) reads the instance count R back to the host once per
# forward, as upstream does, to size the binning scratch.  It keeps ONE thing between calls: the status word of the last
# exact forward, which the next forward's count read-back collects at its own synchronisation (`_State.exact_pending`), so
# that a radix-sort look-back timeout -- the one failure of the exact path -- is raised one forward later instead of never.
#
# Capacity mode is opt-in (`set_sync_free(True)`, or implied by a hipGraph capture): the scratch is sized from the LAST count
# seen for the same (P, W, H) times a headroom factor, the kernels read the live count on the device, and nothing in forward +
# backward synchronises the host -- which is what lets a whole tracking / mapping iteration be captured in a hipGraph.  An
# overflow (R > capacity) drops instances and sets a device flag; `check_overflow()` (one sync, e.g. next to the convergence
# test of the pose step) reports it and raises the capacity so the caller can redo the iteration.  That mode needs memory
# across calls by construction (the hints); it is bounded: at most HINTS_MAX shapes, at most PENDING_MAX unread flags.
_sync_free = {"enabled": False, "headroom": 1.5}
HINTS_MAX, PENDING_MAX = 64, 1024


class _State:
    capacity_hint: dict = {}
    pending: list = []            # (key, flag) of capacity-mode forwards not yet checked
    graph: list = []              # flags of forwards recorded inside a hipGraph capture: re-checked on every call
    exact_pending = None          # (key, flag, device index, raw stream) of the last exact forward, or None
    exact_other: list = []        # the same of exact forwards whose successor ran on ANOTHER stream: only check_overflow() reads them
    exact_failed = 0              # status bits collected from earlier exact forwards, raised by the next forward / check


_capacity_hint = _State.capacity_hint
_pending_overflow = _State.pending
_graph_overflow = _State.graph


def _remember_hint(key, R):
    h = _State.capacity_hint
    if key not in h and len(h) >= HINTS_MAX:
        h.pop(next(iter(h)))
    h[key] = R


def set_sync_free(enabled: bool, headroom: float = 1.5):
    _sync_free["enabled"], _sync_free["headroom"] = bool(enabled), float(headroom)


def sync_free_enabled() -> bool:
    return bool(_sync_free["enabled"])


# MGS_FLAG_EXCLUSIVE_DEVICE (include/monogs_raster.h): the caller vouches that nothing else runs on the device beside the
# forwards issued while this is on -- one process, one stream.  Off by default: MonoGS shares one GPU between three processes.
_call_flags = {"exclusive": False}


@contextlib.contextmanager
def exclusive_device(on: bool = True):
    """``with exclusive_device():`` forwards issued (or captured) inside may assume an otherwise idle device: the small radix
    sorts skip their ticket atomics.  For single-stream loops of a process that owns the GPU (the harness's tracking replays)."""
    prev, _call_flags["exclusive"] = _call_flags["exclusive"], bool(on)
    try:
        yield
    finally:
        _call_flags["exclusive"] = prev


STATUS_CAPACITY_OVERFLOW, STATUS_DEPTH_SORT_TIMEOUT, STATUS_TILE_SORT_TIMEOUT = 1, 2, 4     # MGS_STATUS_* (monogs_raster.h)


def _raise_sort_failure(bits):
    which = [n for b, n in ((STATUS_DEPTH_SORT_TIMEOUT, "depth sort"), (STATUS_TILE_SORT_TIMEOUT, "tile sort")) if bits & b]
    raise RuntimeError(f"rasteriser: a look-back spin of the {' and the '.join(which)} timed out; "
                       "the renders since the last check are invalid")


def _drain_exact_other(n: int) -> None:
    """Nobody calls check_overflow(): read the oldest ``n`` status words of exact forwards that ran on other streams (synchronising
    the devices they ran on) and keep their sort-timeout bits in ``exact_failed``, which the next forward raises -- the list
    stays bounded without losing the one signal it exists to deliver."""
    old, _State.exact_other = _State.exact_other[:n], _State.exact_other[n:]
    for d in {e[2] for e in old}:
        torch.cuda.synchronize(d)
    for e in old:
        _State.exact_failed |= int(e[1].item()) & (STATUS_DEPTH_SORT_TIMEOUT | STATUS_TILE_SORT_TIMEOUT)


def check_overflow() -> bool:
    """Reads the status words of the forwards issued since the last call (synchronises).  True if a capacity-mode
    forward dropped instances -- the capacity hints of the offending shapes are doubled, redo the iteration.  Raises
    if a radix-sort look-back timed out in any forward (exact or capacity mode): its blend order, hence its images
    and gradients, are invalid."""
    hit, sort_fail = False, _State.exact_failed
    _State.exact_failed = 0
    exact = _State.exact_other + ([_State.exact_pending] if _State.exact_pending is not None else [])
    _State.exact_pending = None
    _State.exact_other = []
    # a status word is written by its forward's kernels on the stream (and device) that forward ran on; `.item()` only waits
    # for the CURRENT stream, so drain the devices involved first
    for d in {e[2] for e in exact} | {f.device.index for _, f in _State.pending + _State.graph}:
        torch.cuda.synchronize(d)
    todo = _State.pending + _State.graph + [(e[0], e[1]) for e in exact]
    for key, flag in todo:
        v = int(flag.item())
        if v & STATUS_CAPACITY_OVERFLOW:
            hit = True
            _remember_hint(key, max(2 * _State.capacity_hint.get(key, 1), 1024))
        sort_fail |= v & (STATUS_DEPTH_SORT_TIMEOUT | STATUS_TILE_SORT_TIMEOUT)
    _State.pending.clear()
    if sort_fail:
        _raise_sort_failure(sort_fail)
    return hit


def clear_graph_flags():
    """Forget the overflow flags of captured graphs (call when those graphs are destroyed)."""
    _State.graph.clear()


# ---- optional per-stage timing (bench.py) --------------------------------------------------
_timing_sink: Optional[list] = None


@contextlib.contextmanager
def collect_timing():
    """Within the block every forward/backward appends a dict of per-stage device milliseconds
    (HIP events on the launch stream, see mgs_timing) to the yielded list.  Synchronises."""
    global _timing_sink
    prev, _timing_sink = _timing_sink, []
    try:
        yield _timing_sink
    finally:
        _timing_sink = prev


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _al(n: int) -> int:
    return (int(n) + 255) // 256 * 256


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor (got device {t.device}); "
                           "the rasteriser has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


def _camera(rs: GaussianRasterizationSettings, sh_coeffs: int, keep: list, scale_dim: int = 3) -> _lib.MgsCamera:
    cam = _lib.MgsCamera()
    cam.scale_dim = int(scale_dim)
    cam.flags = 1 if _call_flags["exclusive"] else 0            # MGS_FLAG_EXCLUSIVE_DEVICE
    cam.image_height, cam.image_width = int(rs.image_height), int(rs.image_width)
    cam.tanfovx, cam.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
    cam.scale_modifier = float(rs.scale_modifier)
    cam.sh_degree, cam.sh_coeffs = int(rs.sh_degree), int(sh_coeffs)
    for field in ("bg", "viewmatrix", "projmatrix", "projmatrix_raw", "campos"):
        t = _f32(getattr(rs, field).detach(), f"raster_settings.{field}")
        keep.append(t)
        setattr(cam, field, t.data_ptr())
    return cam


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class _device_guard:
    """``with _device_guard(dev)`` for the common case.  The stock context manager spends ~45 us per use in this
    torch build (``_get_device_index`` consults the environment on entry and exit); five uses per render + loss +
    backward were ~0.2 ms of host time per iteration (tools/host_overhead.py).  Here nothing happens unless ``dev``
    is not the current device."""
    __slots__ = ("idx", "prev")

    def __init__(self, dev):
        idx = getattr(dev, "index", dev)
        self.idx = torch._C._cuda_getDevice() if idx is None else int(idx)

    def __enter__(self):
        self.prev = torch._C._cuda_getDevice()
        if self.prev != self.idx:
            torch._C._cuda_setDevice(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch._C._cuda_setDevice(self.prev)
        return False


def _stream():
    """Raw handle of the caller's current stream.  torch.cuda.current_stream() costs ~40 us per call in this torch
    build (environment lookups inside _get_device_index) -- four calls per render + backward; the C-level accessor
    is ~100x cheaper (tools/host_overhead.py)."""
    if _raw_stream is not None:
        return _raw_stream(torch._C._cuda_getDevice())
    return torch.cuda.current_stream().cuda_stream


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                theta, rho, raster_settings):
        lib = _lib.load()
        rs = raster_settings
        means3D = _f32(means3D.detach(), "means3D")
        P = means3D.shape[0]
        dev = means3D.device
        opt = lambda t, n: _f32(t.detach(), n) if (t is not None and t.numel() > 0) else None  # noqa: E731
        sh_ = opt(sh, "shs")
        col_ = opt(colors_precomp, "colors_precomp")
        opac_ = _f32(opacities.detach(), "opacities")
        sc_ = opt(scales, "scales")
        rot_ = opt(rotations, "rotations")
        cov_ = opt(cov3Ds_precomp, "cov3D_precomp")
        M = 0 if sh_ is None else int(sh_.shape[1])
        H, W = int(rs.image_height), int(rs.image_width)

        with _device_guard(dev):
            keep = []
            scale_dim = int(sc_.shape[1]) if sc_ is not None else 3     # [P,1]: isotropic, expanded inside the kernels
            if scale_dim not in (1, 3):
                raise Exception("scales must be [P,3] (or [P,1] for an isotropic map)")
            cam = _camera(rs, M, keep, scale_dim)
            timing = _lib.MgsTiming() if _timing_sink is not None else None
            tref = C.byref(timing) if timing is not None else None
            u8 = dict(dtype=torch.uint8, device=dev)
            # ONE allocation for everything the backward needs back (geometry + image scratch and the scratch of the backward
            # itself), one for the three images and one for the two per-Gaussian integer results: ten torch.empty per forward
            # were ~35 us of host time at SLAM sizes.  LIFETIME: the arena (~250 B per Gaussian + 8 B per pixel) lives as long as
            # the autograd graph of this forward (it is what the backward reads) -- not as long as any OUTPUT: `radii` and
            # `n_touched` are views of their own 8 P-byte tensor, which a caller may keep (WindowMapper does, per keyframe;
            # MonoGS keeps `radii` in render_pkg) without pinning the scratch.  The three images share one 20 HW-byte tensor.
            want_bwd = P > 0 and any(ctx.needs_input_grad)
            n_geom, n_img = _al(lib.mgs_geometry_bytes(P)), _al(lib.mgs_image_bytes(W, H))
            n_bwd = _al(lib.mgs_backward_bytes(P)) if want_bwd else 0
            arena = torch.empty(n_geom + n_img + n_bwd + 256, **u8)
            base = arena.data_ptr()
            off0 = (-base) % 256                                    # the carving functions expect 256-byte-aligned bases
            geom_p, img_p = base + off0, base + off0 + n_geom
            o_r = off0 + n_geom + n_img
            ri = torch.empty(2, P, dtype=torch.int32, device=dev)
            radii, n_touched = ri[0], ri[1]
            # The scratch of the backward that will follow is handed to the forward: its per-Gaussian kernel clears the
            # gradient lines of the visible Gaussians (and the pose part) on the way, and the backward starts with no
            # clearing launch and no 64 B x P fill.
            ctx.scratch = arena[o_r:o_r + n_bwd] if want_bwd else None
            bwd_p = base + o_r if want_bwd else None
            ctx.scratch_used = False
            out5 = torch.empty(5, H, W, dtype=torch.float32, device=dev)
            color, depth, opacity = out5[0:3], out5[3:4], out5[4:5]
            status = torch.empty(1, dtype=torch.int32, device=dev) if P > 0 else None      # this forward's MGS_STATUS_* word
            key = (P, W, H)
            capturing = torch.cuda.is_current_stream_capturing()
            hint = _State.capacity_hint.get(key)
            if capturing and hint is None:
                raise RuntimeError("graph capture needs a capacity hint: run one eager forward with the same "
                                   "(P, W, H) first")
            if _State.exact_failed and not capturing:
                bits, _State.exact_failed = _State.exact_failed, 0
                _raise_sort_failure(bits)
            if (capturing or _sync_free["enabled"]) and hint is not None and P > 0:
                # ---- capacity mode: no read-back, no stream sync, one crossing of the FFI boundary
                R = max(int(hint * _sync_free["headroom"]) + 4096, 4096)
                binning = torch.empty(lib.mgs_binning_bytes(R, W, H), **u8)
                _lib.check(lib.mgs_forward_capacity(
                    C.byref(cam), P, _ptr(means3D), _ptr(sh_), _ptr(col_), _ptr(opac_), _ptr(sc_), _ptr(rot_), _ptr(cov_),
                    geom_p, radii.data_ptr(), bwd_p, R, binning.data_ptr(), img_p, color.data_ptr(), depth.data_ptr(),
                    opacity.data_ptr(), n_touched.data_ptr(), status.data_ptr(), tref, _stream()), "mgs_forward_capacity")
                (_State.graph if capturing else _State.pending).append((key, status))
                if len(_State.pending) > PENDING_MAX:      # nobody is checking: keep the list bounded
                    del _State.pending[:PENDING_MAX // 2]
                ctx.overflow = status
            else:
                num_rendered, prev_bits = C.c_uint64(0), C.c_uint32(0)
                prev = _State.exact_pending
                here = (torch._C._cuda_getDevice(), _stream())
                if prev is not None and (prev[2], prev[3]) != here:
                    # the earlier forward ran on another stream / device: this stream's read-back is not ordered behind its
                    # kernels (the header's contract is "an EARLIER forward on this stream"), so its word waits for check_overflow()
                    _State.exact_other.append(prev)
                    if len(_State.exact_other) > PENDING_MAX:
                        _drain_exact_other(PENDING_MAX // 2)      # (read, never dropped: a sort timeout must not pass unseen)
                    prev = None
                _lib.check(lib.mgs_forward_preprocess(
                    C.byref(cam), P, _ptr(means3D), _ptr(sh_), _ptr(col_), _ptr(opac_), _ptr(sc_), _ptr(rot_),
                    _ptr(cov_), geom_p, radii.data_ptr(), bwd_p, C.byref(num_rendered),
                    prev[1].data_ptr() if prev is not None else None, C.byref(prev_bits) if prev is not None else None,
                    tref, _stream()), "mgs_forward_preprocess")
                _State.exact_pending = None
                if prev_bits.value & (STATUS_DEPTH_SORT_TIMEOUT | STATUS_TILE_SORT_TIMEOUT):
                    _raise_sort_failure(prev_bits.value)       # the PREVIOUS exact forward's sort timed out: never silent
                R = int(num_rendered.value)
                _remember_hint(key, R)
                binning = torch.empty(lib.mgs_binning_bytes(R, W, H), **u8)
                _lib.check(lib.mgs_forward_render(
                    C.byref(cam), P, R, geom_p, binning.data_ptr(), img_p, color.data_ptr(),
                    depth.data_ptr(), opacity.data_ptr(), n_touched.data_ptr(), _ptr(status), tref, _stream()),
                    "mgs_forward_render")
                if status is not None:      # (the depth sort's flag was already checked at the count read-back)
                    _State.exact_pending = (key, status, here[0], here[1])
                ctx.overflow = None
            if rs.debug:                    # upstream's debug flag: synchronise and check right after the forward
                check_overflow()
            if timing is not None:
                d = timing.as_dict()
                d.update(kind="forward", num_rendered=R, P=P)
                _timing_sink.append(d)

        ctx.cam, ctx.keep = cam, keep          # the backward reuses the camera block (and keeps its tensors alive)
        ctx.geom_off, ctx.img_off = off0, off0 + n_geom
        ctx.raster_settings = rs
        ctx.num_rendered = R
        ctx.sh_coeffs = M
        ctx.scale_dim = scale_dim
        ctx.has = (sh_ is not None, col_ is not None, sc_ is not None, cov_ is not None,
                   theta is not None, rho is not None)
        dummy = torch.empty(0, device=dev)
        ctx.save_for_backward(means3D, sh_ if sh_ is not None else dummy, col_ if col_ is not None else dummy,
                              opac_, sc_ if sc_ is not None else dummy, rot_ if rot_ is not None else dummy,
                              cov_ if cov_ is not None else dummy, radii, arena, binning)
        ctx.mark_non_differentiable(radii, n_touched)
        ctx.set_materialize_grads(False)     # unused output gradients arrive as None instead of three zero-fill kernels
        return color, radii, depth, opacity, n_touched

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_opacity, grad_n_touched):
        # grad_opacity is ignored, as upstream does
        lib = _lib.load()
        rs = ctx.raster_settings
        means3D, sh_, col_, opac_, sc_, rot_, cov_, radii, arena, binning = ctx.saved_tensors
        geom_p, img_p = arena.data_ptr() + ctx.geom_off, arena.data_ptr() + ctx.img_off
        has_sh, has_col, has_sr, has_cov, has_theta, has_rho = ctx.has
        P = means3D.shape[0]
        dev = means3D.device
        H, W = int(rs.image_height), int(rs.image_width)
        need = ctx.needs_input_grad   # means3D, means2D, sh, colors, opacities, scales, rotations, cov3D, theta, rho

        with _device_guard(dev):
            cam = ctx.cam
            f32 = dict(dtype=torch.float32, device=dev)
            g_color = _f32(grad_color, "grad_color") if grad_color is not None else torch.zeros(3, H, W, **f32)
            g_depth = _f32(grad_depth, "grad_depth") if grad_depth is not None else torch.zeros(1, H, W, **f32)
            out = lambda cond, *shape: torch.empty(*shape, **f32) if cond else None  # noqa: E731
            d_means2D = out(need[1], P, 3)
            d_sh = out(need[2] and has_sh, P, max(ctx.sh_coeffs, 1), 3)
            d_cov = out(need[7] and has_cov, P, 6)
            # The gradients of the replicated map parameters (xyz, colour, opacity, scale, rotation) are carved out of ONE
            # allocation, in that order: a keyframe-sharded mapping window can then all-reduce them with a single
            # collective over the flat storage, with no pack / unpack copies (window.GradBucket finds the adjacency).
            widths = [3 if need[0] else 0, 3 if (need[3] and has_col) else 0, 1 if need[4] else 0,
                      ctx.scale_dim if (need[5] and has_sr) else 0, 4 if (need[6] and has_sr) else 0]
            flat = torch.empty(P * sum(widths), **f32) if sum(widths) else None
            views, o = [], 0
            for w in widths:
                views.append(flat[o:o + P * w].view(P, w) if w else None)
                o += P * w
            d_means3D, d_col, d_opac, d_scales, d_rot = views
            want_tau = (need[8] and has_theta) or (need[9] and has_rho)
            prepared = ctx.scratch is not None and not ctx.scratch_used      # (a second backward through the same forward
            ctx.scratch_used = True                                          #  clears a fresh scratch with a launch)
            scratch = ctx.scratch if prepared else torch.empty(lib.mgs_backward_bytes(P), dtype=torch.uint8, device=dev)
            if want_tau and prepared:       # the six floats live inside the prepared scratch (the forward cleared them)
                off = int(lib.mgs_backward_tau(scratch.data_ptr(), P)) - scratch.data_ptr()
                d_tau = scratch[off:off + 24].view(torch.float32)
            else:
                d_tau = torch.empty(6, **f32) if want_tau else None
            timing = _lib.MgsTiming() if _timing_sink is not None else None
            tref = C.byref(timing) if timing is not None else None
            _lib.check(lib.mgs_backward(
                C.byref(cam), P, ctx.num_rendered,
                _ptr(means3D), _ptr(sh_) if has_sh else None, _ptr(col_) if has_col else None, _ptr(opac_),
                _ptr(sc_) if has_sr else None, _ptr(rot_) if has_sr else None, _ptr(cov_) if has_cov else None,
                radii.data_ptr(), geom_p, binning.data_ptr(), img_p,
                g_color.data_ptr(), g_depth.data_ptr(),
                _ptr(d_means2D), _ptr(d_col), _ptr(d_opac), _ptr(d_means3D), _ptr(d_cov), _ptr(d_sh),
                _ptr(d_scales), _ptr(d_rot), _ptr(d_tau), scratch.data_ptr(), 1 if prepared else 0, tref, _stream()),
                "mgs_backward")
            if timing is not None:
                d = timing.as_dict()
                d.update(kind="backward", num_rendered=ctx.num_rendered, P=P)
                _timing_sink.append(d)
        # (views of the 6-float result: autograd takes them as .grad without a copy kernel)
        d_theta = d_tau[3:] if (d_tau is not None and need[8] and has_theta) else None
        d_rho = d_tau[:3] if (d_tau is not None and need[9] and has_rho) else None
        return (d_means3D, d_means2D, d_sh, d_col, d_opac, d_scales, d_rot, d_cov, d_theta, d_rho, None)


def debug_blend_stats(color: torch.Tensor) -> dict:
    """Diagnostic: what the blend backward of the forward that produced ``color`` walks, fetches and reduces
    (``mgs_debug_blend_stats``; call before ``backward()`` frees the saved scratch).  Synchronises."""
    fn = color.grad_fn
    if fn is None or not hasattr(fn, "raster_settings"):
        raise RuntimeError("debug_blend_stats needs the colour image of a differentiable rasteriser forward")
    lib = _lib.load()
    means3D, _, _, _, _, _, _, _, arena, binning = fn.saved_tensors
    cam = fn.cam
    out = torch.zeros(24, dtype=torch.int64, device=means3D.device)      # MGS_BLEND_STATS_WORDS
    with _device_guard(means3D.device):
        _lib.check(lib.mgs_debug_blend_stats(C.byref(cam), means3D.shape[0], fn.num_rendered,
                                             arena.data_ptr() + fn.geom_off, binning.data_ptr(),
                                             arena.data_ptr() + fn.img_off, out.data_ptr(), _stream()),
                   "mgs_debug_blend_stats")
    v = out.tolist()
    groups = {}
    for d, name in enumerate(("halves_8x4", "halves_4x8", "blocks_4x4", "strips_8x2", "blocks_4x2")):
        groups[name] = dict(trips_paired_per_step=v[8 + 3 * d], rows=v[9 + 3 * d], trips_own_lists=v[10 + 3 * d])
    return dict(steps=v[0], survivors=v[1], active_survivors=v[2], active_pairs=v[3], inactive_by_depth_order=v[4],
                active_le2=v[5], active_le4=v[6], active_le8=v[7], group_streams=groups)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        theta, rho, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, theta, rho, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """bool[P]: inside the view frustum's near side (view-space z > 0.2)."""
        lib = _lib.load()
        rs = self.raster_settings
        with torch.no_grad():
            pos = _f32(positions, "positions")
            P = pos.shape[0]
            vis = torch.zeros(P, dtype=torch.uint8, device=pos.device)
            with _device_guard(pos.device):
                vm = _f32(rs.viewmatrix, "viewmatrix")
                pm = _f32(rs.projmatrix, "projmatrix")
                _lib.check(lib.mgs_mark_visible(P, pos.data_ptr(), vm.data_ptr(), pm.data_ptr(), vis.data_ptr(),
                                                _stream()), "mgs_mark_visible")
        return vis.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, theta, rho, rs)
