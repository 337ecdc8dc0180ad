"""Does the pipelined clip driver lose its overlap when the process uses more HIP streams than there are hardware queues?
(ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default, in order of first use.)  Times the eager clip
driver (default stream + recurrence stream + 2 side streams = 4) with 0 / 1 / 2 additional streams that have each run one
kernel.  Run once as is and once with GPU_MAX_HW_QUEUES=8 in the environment.  GPU box: python tools/hw_queue_probe.py"""
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
ops.set_autotune(True)
IB = synth.synth_lab(2, H, W).to(dev)
fr = [synth.synth_lab(1000 + i, H, W).to(dev) for i in range(16)]
cc = ClipColorizer(*nets)
cc.set_exemplar(IB)
cc.clip(fr[:6], lookahead=2)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(unset: 4)"))
extra = []
for n_extra in (0, 1, 2, 4):
    while len(extra) < n_extra:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(16, device=dev).add_(1)      # the stream has run a kernel: it owns / shares a hardware queue now
        s.synchronize()
        extra.append(s)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cc.clip(fr, lookahead=2)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / len(fr))
    print(f"eager clip driver, {n_extra} extra stream(s) in the process: {best * 1e3:.3f} ms/frame ({1 / best:.1f} frames/s)", flush=True)
