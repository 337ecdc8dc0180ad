"""Where is the tail of frame 1000's error field, and whose is it?  |ab - fp64 oracle| at 216x384 (plain seed-0 weights, T = 1e-10)
for the reference arithmetic (oracle, torch CPU fp32) at 16 and at 4 ATen threads, and for the HIP path with / without the
weights-in-registers engine and the autotuner: q99.9, max, mean and the five largest errors with their positions.
-> profiles/r06_parity_hotspot.txt (GPU box)."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from dvc_amd import ops, synth
from dvc_amd.frame import ClipColorizer
from models.ColorVidNet import ColorVidNet
from models.NonlocalNet import VGG19_pytorch, WarpNet
from oracle import dvc_oracle as O
H, W, T = 216, 384, 1e-10
dev = torch.device("cuda")
with contextlib.redirect_stdout(io.StringIO()):
    nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
sd = (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))
for m, s in zip(nets, sd):
    m.load_state_dict(s); m.eval().to(dev)
IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W); fr = synth.synth_lab(1000, H, W); z = torch.zeros_like(fr)
torch.set_num_threads(16); torch.set_flush_denormal(True)
sd64 = tuple(O.to_dtype(s, torch.float64) for s in sd)
with torch.no_grad():
    a32 = O.frame_colorization(fr, IB, z, O.exemplar_features(IB, sd[0]), *sd, temperature=T)[0]
    a64 = O.frame_colorization(fr.double(), IB.double(), z.double(), O.exemplar_features(IB.double(), sd64[0]), *sd64, temperature=T)[0]
    torch.set_num_threads(4)
    a32b = O.frame_colorization(fr, IB, z, O.exemplar_features(IB, sd[0]), *sd, temperature=T)[0]
ec = (a32.double() - a64).abs()[0]; ecb = (a32b.double() - a64).abs()[0]
def top(e, name):
    v, i = e.flatten().topk(5)
    print(name, "q999 %.2e max %.2e mean %.2e" % (np.quantile(e.numpy(), 0.999), e.max(), e.mean()), [(int(j) // (H * W), (int(j) % (H * W)) // W, int(j) % W, "%.1e" % x) for x, j in zip(v.tolist(), i.tolist())])
top(ec, "cpu32 16 threads"); top(ecb, "cpu32  4 threads")
for name, ws, tune in (("gpu ws", True, False), ("gpu ws+tune", True, True), ("gpu nows", False, False), ("gpu nows+tune", False, True)):
    ops.set_ws_conv(ws); ops.set_autotune(tune)
    cc = ClipColorizer(*nets, temperature=T, graph=False); cc.set_exemplar(IB.to(dev))
    ab, _ = cc.frame(fr.to(dev), z.to(dev))
    top((ab.double().cpu() - a64).abs()[0], name)
