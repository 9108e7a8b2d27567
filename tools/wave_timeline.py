"""Occupancy over time of the two blend kernels, from per-wave {start, end} stamps of a DIAGNOSTIC build.

    tools/build_variant.sh trace blend.hip -DBS_TRACE          # here
    MGS_LIB_PATH=$PWD/monogs_amd/lib/variants/libmgs_trace.so python tools/wave_timeline.py   # on the GPU box

For the C5 scene (and 100 k / VGA): how many waves are resident in each tenth of the kernel's span, how long a wave lives, how the
lifetimes of the four waves of a workgroup differ, how evenly the work (survivors) is spread over the SIMDs, and how much of the
span is ramp-up and tail -- the 52 us (backward) / 28 us (forward) that tools/tail_probe.py finds independent of the image size.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monogs_amd import _lib  # noqa: E402
from monogs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from monogs_amd.synthetic import make_scene, scene_settings  # noqa: E402


def analyse(name, tr):
    tr = tr.reshape(-1, 4)
    ok = tr[:, 1] > 0
    t0, t1, hw, n = tr[ok, 0].astype(np.float64), tr[ok, 1].astype(np.float64), tr[ok, 2], tr[ok, 3].astype(np.float64)
    base = t0.min()
    t0, t1 = (t0 - base) * 0.01, (t1 - base) * 0.01        # microseconds (100 MHz stamps)
    span = t1.max()
    life = t1 - t0
    hwid = (hw & 0xFFFFFFFF).astype(np.int64)
    xcc = (hw >> 32).astype(np.int64) & 0xF
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    simd_key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    nsimd = len(np.unique(simd_key))
    out = {"kernel": name, "waves": int(ok.sum()), "span_us": round(float(span), 1), "simds_seen": int(nsimd),
           "wave_life_us": {k: round(float(np.percentile(life, p)), 1) for k, p in (("p10", 10), ("p50", 50), ("p90", 90), ("max", 100))}}
    # resident waves per SIMD slot (8 per SIMD) in each tenth of the span
    edges = np.linspace(0, span, 11)
    occ = []
    for a, b in zip(edges[:-1], edges[1:]):
        ov = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None).sum() / (b - a)
        occ.append(round(float(ov / (nsimd * 8)), 3))
    out["occupancy_by_tenth_of_span"] = occ
    out["mean_occupancy"] = round(float(life.sum() / (span * nsimd * 8)), 3)
    # when does the last wave START, when is half of the SIMDs out of work
    out["last_wave_starts_at_us"] = round(float(t0.max()), 1)
    last_end = np.zeros(nsimd)
    keys, inv = np.unique(simd_key, return_inverse=True)
    np.maximum.at(last_end, inv, t1)
    out["simd_finish_us"] = {k: round(float(np.percentile(last_end, p)), 1) for k, p in (("p10", 10), ("p50", 50), ("p90", 90), ("max", 100))}
    work = np.zeros(nsimd)
    np.add.at(work, inv, n)
    out["work_per_simd_rel"] = {k: round(float(np.percentile(work, p) / work.mean()), 3) for k, p in (("p1", 1), ("p50", 50), ("p99", 99), ("max", 100))}
    # spread of the four waves of a workgroup
    if ok.all():
        l4 = life.reshape(-1, 4)
        out["wg_wave_life_min_over_max_p50"] = round(float(np.median(l4.min(1) / l4.max(1))), 3)
    # time at which the resident count first reaches 90 % of its plateau
    ts = np.linspace(0, span, 401)
    res = np.array([((t0 <= t) & (t1 > t)).sum() for t in ts]) / (nsimd * 8)
    plateau = np.percentile(res, 90)
    up = ts[np.argmax(res >= 0.9 * plateau)]
    down = ts[len(ts) - 1 - np.argmax(res[::-1] >= 0.9 * plateau)]
    out["plateau_occupancy"] = round(float(plateau), 3)
    out["ramp_up_us"] = round(float(up), 1)
    out["tail_us"] = round(float(span - down), 1)
    print(json.dumps(out), flush=True)


def run(P, intr):
    dev = torch.device("cuda", 0)
    lib = C.CDLL(_lib.LIB_PATH)
    sc = make_scene(P, intr, seed=2)
    st = scene_settings(sc, GaussianRasterizationSettings, device=dev)
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
    xyz, rgb, opac, scaling, rot = leaf(sc.means3D), leaf(sc.colors), leaf(sc.opacities), leaf(sc.scales), leaf(sc.rotations)
    gc, gd = sc.grad_color.to(dev), sc.grad_depth.to(dev)
    r = GaussianRasterizer(st)
    tiles = ((st.image_width + 15) // 16) * ((st.image_height + 15) // 16)
    tf = torch.zeros(tiles * 16, dtype=torch.int64, device=dev)
    tb = torch.zeros(tiles * 16, dtype=torch.int64, device=dev)
    for i in range(6):
        for p in (xyz, rgb, opac, scaling, rot):
            p.grad = None
        if i == 5:
            lib.mgs_trace_set_blend_buffers(C.c_void_p(tf.data_ptr()), C.c_void_p(tb.data_ptr()))
        m2 = torch.zeros_like(xyz, requires_grad=True)
        out = r(means3D=xyz, means2D=m2, opacities=opac, colors_precomp=rgb, scales=scaling, rotations=rot)
        torch.autograd.backward([out[0], out[2]], [gc, gd])
    torch.cuda.synchronize()
    lib.mgs_trace_set_blend_buffers(None, None)
    print(f"# {P} Gaussians, {st.image_width}x{st.image_height}, {tiles} tiles")
    analyse("blend_forward", tf.cpu().numpy().astype(np.uint64))
    analyse("blend_backward", tb.cpu().numpy().astype(np.uint64))


if __name__ == "__main__":
    assert "trace" in _lib.LIB_PATH, "run with MGS_LIB_PATH pointing at the BS_TRACE variant"
    run(2_000_000, "davis_1080p")
    run(100_000, "fr3_office")
