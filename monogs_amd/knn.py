"""``distCUDA2`` -- host side of the simple-knn drop-in.

Call site: /root/reference/gaussian_splatting/scene/gaussian_model.py:294-302 (one positional
float32 CUDA tensor [P,3] -> float32 [P], mean squared distance to the 3 nearest other points).
"""
from __future__ import annotations

import torch

from . import _lib
from .rasterizer import _stream, _device_guard


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 expects a CUDA/HIP tensor; there is no CPU path")
    pts = points.detach().to(torch.float32).contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError(f"distCUDA2 expects [P,3], got {tuple(pts.shape)}")
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    with _device_guard(pts.device):
        scratch = torch.empty(lib.mgs_knn_scratch_bytes(P), dtype=torch.uint8, device=pts.device)
        _lib.check(lib.mgs_dist2_knn(P, pts.data_ptr(), out.data_ptr(), scratch.data_ptr(),
                                     _stream()), "mgs_dist2_knn")
    return out
