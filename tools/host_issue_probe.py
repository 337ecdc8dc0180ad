"""How long does the HOST need to issue one frame's launches?  (Is the clip driver limited by the launch thread?)
Issues 3 frames of the per-frame loop into an idle stream without synchronising and times the host side alone, then the
same with the GPU time included.  GPU box: python tools/host_issue_probe.py"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")); sys.path.insert(0, ROOT)
import torch
torch.set_num_threads(8)
from dvc_amd import ops, synth
from dvc_amd.frame import ClipColorizer
from models.ColorVidNet import ColorVidNet
from models.NonlocalNet import VGG19_pytorch, WarpNet

H, W = 216, 384
dev = torch.device("cuda")
with contextlib.redirect_stdout(io.StringIO()):
    nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
for m, s in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
    m.load_state_dict(s); m.eval().to(dev)
cc = ClipColorizer(*nets)
cc.set_exemplar(synth.synth_lab(2, H, W).to(dev))
fr = [synth.synth_lab(1000 + i, H, W).to(dev) for i in range(8)]
last = torch.zeros_like(fr[0])
for i in range(8):
    ab, _ = cc.frame(fr[i], last)
torch.cuda.synchronize()
ops.conv_record = []
cc.frame(fr[0], last)
n_conv = len(ops.conv_record); ops.conv_record = None
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(3):
        ab, _ = cc.frame(fr[i], last)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"3 frames: host issue {(t1 - t0) / 3 * 1e3:.2f} ms/frame, until the GPU is done {(t2 - t0) / 3 * 1e3:.2f} ms/frame "
          f"({n_conv} convolution launches per frame)", flush=True)
# ---- the same through captured launch sequences (hipGraph replay, dvc_amd/graph.py): per-frame API and the clip driver
ops.set_autotune(True)
cg = ClipColorizer(*nets, graph=True)
cg.set_exemplar(synth.synth_lab(2, H, W).to(dev))
for i in range(4):
    cg.frame(fr[i], last)
cg.clip(fr, lookahead=2)
cc.clip(fr, lookahead=2)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(3):
        ab, _ = cg.frame(fr[i], last)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"graph replay, 3 per-frame calls: host issue {(t1 - t0) / 3 * 1e3:.3f} ms/frame, until the GPU is done "
          f"{(t2 - t0) / 3 * 1e3:.2f} ms/frame", flush=True)
for name, drv in (("eager", cc), ("graph", cg)):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv.clip(fr, lookahead=2)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"clip driver ({name}), 8 frames, look-ahead 2: host issue {(t1 - t0) / 8 * 1e3:.3f} ms/frame, until the GPU is done "
              f"{(t2 - t0) / 8 * 1e3:.2f} ms/frame", flush=True)
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for i in range(3):
    ab, _ = cc.frame(fr[i], last)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
