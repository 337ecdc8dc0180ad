"""Do shorter-lived background workgroups help the critical path?  The look-ahead front ends run on low-priority streams; their
workgroups are not preemptible, so a ColorVidNet-chain kernel that becomes ready has to wait for them to drain.  Here the
front-end Winograd launches are planned with twice the split over input channels (dvc_debug_conv_variant(64), debug build)
— half as long per workgroup, more partial-sum traffic — while the chain keeps its plan.  GPU box: python tools/bg_split_probe.py"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("DVC_DEBUG_LIB", "1")
import torch
torch.set_num_threads(8)
from dvc_amd import _lib, frame, ops, synth
from dvc_amd.frame import ClipColorizer
from models.ColorVidNet import ColorVidNet
from models.NonlocalNet import VGG19_pytorch, WarpNet

H, W = 216, 384
dev = torch.device("cuda")
lib = _lib.load()
with contextlib.redirect_stdout(io.StringIO()):
    nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
for m, s in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
    m.load_state_dict(s); m.eval().to(dev)
ops.set_autotune(True)
IB = synth.synth_lab(2, H, W).to(dev)
fr = [synth.synth_lab(1000 + i, H, W).to(dev) for i in range(16)]
cc = ClipColorizer(*nets)
cc.set_exemplar(IB)
real_warp_color = frame.warp_color
mode = {"v": 0}


def warp_color_bg(*a, **k):
    lib.dvc_debug_conv_variant(mode["v"])
    try:
        return real_warp_color(*a, **k)
    finally:
        lib.dvc_debug_conv_variant(0)


frame.warp_color = warp_color_bg
real_col_forward = type(nets[2]).forward
cmode = {"v": 0}


def col_forward(self, x):
    lib.dvc_debug_conv_variant(cmode["v"])
    try:
        return real_col_forward(self, x)
    finally:
        lib.dvc_debug_conv_variant(0)


type(nets[2]).forward = col_forward
for name, v, cv in (("library's plan everywhere", 0, 0), ("front-end Winograd launches padded to one workgroup per CU", 512, 0), ("front ends with twice the split", 64, 0), ("front ends with half the split", 128, 0),
                    ("front ends without split", 256, 0), ("ColorVidNet chain with half the split", 0, 128),
                    ("ColorVidNet chain without split", 0, 256), ("both with half the split", 128, 128), ("ColorVidNet chain with twice the split", 0, 64),
                    ("chain twice, front ends half", 128, 64), ("library's plan again", 0, 0)):
    mode["v"] = v
    cmode["v"] = cv
    cc.clip(fr[:6], lookahead=2)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cc.clip(fr, lookahead=2)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / len(fr))
    print(f"{name:40s}: {best * 1e3:.3f} ms/frame ({1 / best:.1f} frames/s)", flush=True)
