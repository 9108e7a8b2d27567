"""Test/diagnostic access to the binning tables the forward leaves in its scratch buffers.
Calls the C ABI directly (include/monogs_raster.h) and decodes the documented scratch layout
(256-byte aligned sub-buffers in declaration order, see csrc/api.hip)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .rasterizer import _camera, _f32, _ptr, _stream, _device_guard


def _au(v, a=256):
    return (v + a - 1) // a * a


def forward_tables(rs, means3D, opacities, colors_precomp=None, shs=None, scales=None, rotations=None,
                   cov3D_precomp=None, between=None):
    """Run the HIP forward and return outputs plus the per-tile tables as torch tensors.  ``between``: called after the
    first stage (per-Gaussian kernel, depth sort, scan) has finished and before the second (duplicate, tile sort, blend) starts."""
    lib = _lib.load()
    dev = means3D.device
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    c = lambda t, n: None if t is None else _f32(t.detach(), n)  # noqa: E731
    means3D, opacities = c(means3D, "means3D"), c(opacities, "opacities")
    colors_precomp, shs, scales = c(colors_precomp, "colors"), c(shs, "shs"), c(scales, "scales")
    rotations, cov3D_precomp = c(rotations, "rotations"), c(cov3D_precomp, "cov3D")
    with _device_guard(dev):
        keep = []
        scale_dim = int(scales.shape[1]) if scales is not None else 3       # [P,1]: isotropic, expanded inside the kernels
        if scale_dim not in (1, 3):
            raise Exception("scales must be [P,3] (or [P,1] for an isotropic map)")
        cam = _camera(rs, 0 if shs is None else shs.shape[1], keep, scale_dim)
        u8 = dict(dtype=torch.uint8, device=dev)
        geom = torch.zeros(lib.mgs_geometry_bytes(P), **u8)
        img = torch.zeros(lib.mgs_image_bytes(W, H), **u8)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        n_touched = torch.empty(P, dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        opacity = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        nr = C.c_uint64(0)
        _lib.check(lib.mgs_forward_preprocess(C.byref(cam), P, _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
                                              _ptr(opacities), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp),
                                              geom.data_ptr(), radii.data_ptr(), None, C.byref(nr), None, None, None, _stream()),
                   "mgs_forward_preprocess")
        R = int(nr.value)
        if between is not None:
            between()
        binning = torch.zeros(lib.mgs_binning_bytes(R, W, H), **u8)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.mgs_forward_render(C.byref(cam), P, R, geom.data_ptr(), binning.data_ptr(), img.data_ptr(),
                                          color.data_ptr(), depth.data_ptr(), opacity.data_ptr(),
                                          n_touched.data_ptr(), status.data_ptr() if P > 0 else None, None, _stream()),
                   "mgs_forward_render")
        torch.cuda.synchronize()

    def view(buf, off, nbytes, dtype):
        base = _au(buf.data_ptr()) - buf.data_ptr()
        return buf[base + off: base + off + nbytes].view(dtype)

    HW = H * W
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    final_T = view(img, 0, HW * 4, torch.float32).reshape(H, W)
    n_contrib = view(img, _au(HW * 4), HW * 4, torch.int32).reshape(H, W)
    ranges = view(img, 2 * _au(HW * 4), ntiles * 8, torch.int32).reshape(ntiles, 2)
    r = max(R, 1)
    # binning scratch: keys_a, keys_b, vals_a, vals_b; an LSD pass per 8 key bits ping-pongs a -> b -> a ...
    tile_bits = max(1, (ntiles - 1).bit_length())
    in_b = ((tile_bits + 7) // 8) % 2 == 1
    # tile id per sorted instance.  Since round 4 the tile sort's final pass writes the per-tile ranges instead of the sorted
    # keys (nobody reads them): the tile of instance i is the tile whose range holds i.
    rg = ranges.long()
    tile_sorted = torch.repeat_interleave(torch.arange(ntiles, device=ranges.device), (rg[:, 1] - rg[:, 0]).clamp_min(0)).to(torch.int32)
    point_list = view(binning, (3 if in_b else 2) * _au(r * 4), R * 4, torch.int32)
    rec = view(geom, 0, P * 64, torch.float32).reshape(P, 16)
    o = _au(P * 64)
    # geometry scratch after rec (GeometryState::carve, csrc/api.hip): depth_key, depth_alt, iota, iota_alt (the depth
    # sort's ping-pong buffers), perm (its result), point_offsets, scan_blocks, clamped, rect (by Gaussian index: consumed
    # by the sort, possibly as its packed payload), rect_sorted {x0 | y0 << 16, w | h << 16} in depth order
    perm = view(geom, o + 4 * _au(P * 4), P * 4, torch.int32)
    o_rect = o + 6 * _au(P * 4) + _au(((P + 2047) // 2048 + 64) * 4) + _au(P * 4)
    wh_sorted = view(geom, o_rect + _au(P * 8), P * 8, torch.int32).reshape(P, 2)[:, 1]
    tiles_touched = torch.zeros(P, dtype=torch.int32, device=geom.device)
    # tiles each Gaussian touches = w * h of its rectangle (scattered back from depth order to Gaussian index)
    tiles_touched[perm.long()] = (wh_sorted & 0xFFFF) * ((wh_sorted >> 16) & 0xFFFF)
    depth_key = rec[:, 11].contiguous().view(torch.int32)          # float32 bits of the view-space depth
    return dict(color=color, depth=depth, opacity=opacity, radii=radii, n_touched=n_touched, num_rendered=R,
                status=int(status.item()),
                final_T=final_T, n_contrib=n_contrib, ranges=ranges, tile_sorted=tile_sorted,
                point_list=point_list, rec=rec, tiles_touched=tiles_touched, depth_key=depth_key, perm=perm)
