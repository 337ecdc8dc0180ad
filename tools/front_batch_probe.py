"""Front end (VGG19 + WarpNet + correlation) of two frames: two single-frame passes against one batch-of-two pass, one
stream (is batching the look-ahead front ends worth building?).  GPU box: python tools/front_batch_probe.py"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")); sys.path.insert(0, ROOT)
import torch
from dvc_amd import ops, synth
from dvc_amd.frame import ClipColorizer, warp_color
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
f = [synth.synth_lab(1000 + i, H, W).to(dev) for i in range(2)]
phi, blab = cc.ex_cache
for nb in (2, 4):
    fb = torch.cat([f[i % 2] for i in range(nb)], 0)
    cache_b = (phi.expand(nb, -1, -1).contiguous(), blab.expand(nb, -1, -1, -1).contiguous())
    IBb = cc.IB_lab.expand(nb, -1, -1, -1).contiguous()
    def single():
        for i in range(nb):
            warp_color(f[i % 2][:, 0:1], cc.IB_lab, None, cc.vgg, cc.warp, cc.col, 0, temperature=1e-10, exemplar_cache=cc.ex_cache)
    def batched():
        warp_color(fb[:, 0:1].contiguous(), IBb, None, cc.vgg, cc.warp, cc.col, 0, temperature=1e-10, exemplar_cache=cache_b)
    def timeit(fn, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for _ in range(5): single(); batched()
    best = {"single": 1e9, "batched": 1e9}
    for r in range(4):
        best["single"] = min(best["single"], timeit(single)); best["batched"] = min(best["batched"], timeit(batched))
    print(f"{nb} frames: {nb} single-frame front ends {best['single']:.0f} us, one batch-of-{nb} front end {best['batched']:.0f} us "
          f"({best['batched'] / best['single'] * 100:.1f} %)", flush=True)
