import torch, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from monogs_amd.debug import forward_tables
from monogs_amd.rasterizer import GaussianRasterizationSettings
from monogs_amd.synthetic import make_scene, scene_settings
DEV = "cuda:0"
for P, intr in ((100000, "fr3_office"), (40000, "fr3_office"), (2000000, "davis_1080p")):
    sc = make_scene(P, intr, seed=1)
    st = scene_settings(sc, GaussianRasterizationSettings, device=DEV)
    dev = lambda x: x.to(DEV)
    t = forward_tables(st, dev(sc.means3D), dev(sc.opacities), colors_precomp=dev(sc.colors), scales=dev(sc.scales), rotations=dev(sc.rotations))
    r = t["ranges"].long()
    ln = (r[:, 1] - r[:, 0]).float()
    nc = t["n_contrib"].float()
    H, W = nc.shape[-2:]
    # per-quadrant max n_contrib (what a wave walks)
    q = nc.reshape(H // 8 if H % 8 == 0 else -1, 8, W // 8, 8) if H % 8 == 0 and W % 8 == 0 else None
    print(P, intr, "R", t["num_rendered"], "tiles", ln.numel(), "len mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (ln.mean(), ln.median(), ln.quantile(0.9), ln.quantile(0.99), ln.max()))
    if q is not None:
        qm = q.amax(dim=(1, 3)).flatten()
        print("   quadrant walk length (max n_contrib): mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (qm.mean(), qm.median(), qm.quantile(0.9), qm.quantile(0.99), qm.max()))
