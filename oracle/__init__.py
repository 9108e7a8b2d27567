"""CPU oracle for the MonoGS rasteriser hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  Nothing under ``monogs_amd/`` imports it; the product
path fails loudly when the HIP library is missing instead of falling back to this code.

PARITY UNPINNED: the reference's rasteriser and simple-knn are un-vendored submodules
(/root/reference/.gitmodules:1-6, both directories empty), and the reference ships no tests,
golden vectors or fixtures for this path.  The oracle is therefore a
restatement of the published algorithm, pinned by
  * the reference helpers that *do* import here (pose retraction, camera matrices, SH
    evaluation; see tests/golden/make_golden.py and tests/test_golden.py),
  * float64 ``torch.autograd.gradcheck`` and central finite differences of the pose Jacobian,
  * known-answer cases (tests/test_oracle_kat.py),
  * the reference's OpenGL viewer shaders, the only splatting code it holds: covariance, EWA projection (1.3 tan(fov)
    clamp, +0.3 low-pass), conic and the per-fragment alpha rule restated from
    /root/reference/viewer/gl_render/shaders/gau_vert.glsl:60-107,149-154 and gau_frag.glsl:20-25
    (tests/test_reference_shader.py).
"""
from .gs_oracle import (  # noqa: F401
    OracleSettings,
    preprocess,
    build_binning,
    rasterize,
    rasterize_autograd,
    se3_exp,
    BLOCK_X,
    BLOCK_Y,
)
from .knn_oracle import dist2_knn  # noqa: F401
