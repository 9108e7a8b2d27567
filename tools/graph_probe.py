"""Diagnostic: which captured content survives eager work between hipGraph replays?  (prints before each step)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from monogs_amd import rasterizer as _r
from monogs_amd.renderer import render
from monogs_amd.slam_harness import make_sequence
from monogs_amd import fused_losses
from monogs_amd.pose_optim import PoseAdam


def say(*a):
    torch.cuda.synchronize()
    print("[probe]", *a, file=sys.stderr, flush=True)


def exercise(name, graph, outs):
    for step in ("replay", "replay", "tolist", "replay", "copy", "replay", "newkernel", "replay"):
        if step == "replay":
            graph.replay()
        elif step == "tolist":
            outs[0].flatten()[:2].tolist()
        elif step == "copy":
            outs[0].copy_(outs[0].clone())
        else:
            (outs[0].double().cumsum(0).to(torch.int16))
        say(name, step, "ok")


dev = "cuda:0"
# V1: torch only
a = torch.randn(4, 4, device=dev)
b = torch.randn(4, 4, device=dev)
o = torch.zeros(4, 4, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    o.copy_((a.unsqueeze(0).bmm(b.unsqueeze(0))).squeeze(0) + 1)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    o.copy_((a.unsqueeze(0).bmm(b.unsqueeze(0))).squeeze(0) + 1)
exercise("V1 torch-only", g, [o])

# scene
frames, intr = make_sequence(2, "fr3_office", 20000, device=dev)
vp = frames[1]
vp.update_RT(frames[0].R_gt.clone(), frames[0].T_gt.clone())
from monogs_amd.synthetic import make_scene
sc = make_scene(20000, "fr3_office", seed=11, near_fraction=0.0, mean_radius_px=9.0, device=dev)
xyz, rot, sca, opa, col = sc.means3D, sc.rotations, sc.scales, sc.opacities, sc.colors
bg = torch.zeros(3, device=dev)


def capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def fwd():
    with torch.no_grad():
        return render(vp, intr, xyz, rot, sca, opa, col, bg)["render"]


g2, out2 = capture(fwd)
exercise("V2 forward", g2, [out2])
_r.check_overflow(); _r.clear_graph_flags(); del g2

opt = PoseAdam(vp, 0.003, 0.001, 0.01)


def fwd_bwd():
    pkg = render(vp, intr, xyz, rot, sca, opa, col, bg)
    opt.zero_grad()
    loss = fused_losses.get_loss_tracking(pkg["render"], pkg["depth"], pkg["opacity"], vp)
    loss.backward()
    return pkg["render"]


g3, out3 = capture(fwd_bwd)
exercise("V3 fwd+loss+bwd", g3, [out3])
_r.check_overflow(); _r.clear_graph_flags(); del g3


def full():
    r = fwd_bwd()
    opt.step_and_retract(sync=False)
    return r


g4, out4 = capture(full)
exercise("V4 full iteration", g4, [out4])
say("all variants ok")
