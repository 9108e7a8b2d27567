"""CPU oracle for ``simple_knn._C.distCUDA2``.  TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Contract (call site /root/reference/gaussian_splatting/scene/gaussian_model.py:294-302;
implementation un-vendored): for every point the MEAN of
the SQUARED Euclidean distances to its 3 nearest OTHER points, exact, float32.  With fewer than
4 points the missing neighbours contribute FLT_MAX terms upstream; the oracle returns +inf there
and tests only use P >= 4.
"""
import numpy as np
import torch


def dist2_knn(points: torch.Tensor) -> torch.Tensor:
    from scipy.spatial import cKDTree

    p = points.detach().cpu().to(torch.float64).numpy()
    n = p.shape[0]
    if n < 4:
        return torch.full((n,), float("inf"), dtype=torch.float32)
    tree = cKDTree(p)
    d, _ = tree.query(p, k=4)           # column 0 is the point itself (distance 0)
    # duplicates: cKDTree may return another coincident point first; distances are what matter
    d2 = np.sort(d, axis=1)[:, 1:4] ** 2
    return torch.from_numpy(d2.mean(axis=1).astype(np.float32))
