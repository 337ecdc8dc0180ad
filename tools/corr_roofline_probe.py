#!/usr/bin/env python
"""The correlation stage alone, on the clip's own operands (theta of frame seed 1005 through VGG19 -> WarpNet -> 1x1 ->
centre / normalise, phi and pooled Lab of exemplar seed 2: what bench.py's roofline leg times), for a rocprofv3 kernel trace
whose per-kernel average is NOT mixed with the launches that share the GPU with two other streams in the clip driver:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/corrprof -o t -- python tools/corr_roofline_probe.py

Prints the HIP-event averages of the three forms: kernel alone (merge deferred to the consumer: what the clip driver
launches), kernel + stand-alone merge, and the consumer's merge + pack launch."""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402

from dvc_amd import ops, synth  # noqa: E402
from dvc_amd.frame import VGG_OUT, ClipColorizer  # noqa: E402
from models.ColorVidNet import ColorVidNet  # noqa: E402
from models.NonlocalNet import VGG19_pytorch, WarpNet  # noqa: E402

H, W = 216, 384
h, w = H // 4, W // 4
P = h * w
FLOPS = 2.0 * P * P * 256 + 2.0 * P * P * 3
dev = torch.device("cuda")
with contextlib.redirect_stdout(io.StringIO()):
    nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
for m, sd in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
    m.load_state_dict(sd)
    m.eval().to(dev)
vgg, warp, col = nets
cc = ClipColorizer(vgg, warp, col, temperature=1e-10)
cc.set_exemplar(synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(dev))
fr = synth.synth_lab(synth.FRAME_SEED0 + 5, H, W).to(dev)
fA = vgg(ops.gray2rgb(fr[:, 0:1]), VGG_OUT)
th = warp.project("theta", warp.features(*ops.channel_l2norm_multi(fA[1:])))
ph, bl4 = cc.ex_cache
bl = bl4.view(1, 3, -1)
last = torch.zeros_like(fr)


def timed(fn, warm=150, reps=200):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


t_k = timed(lambda: ops.corr_fwd(th, ph, bl, 1e-10, h, w, defer_merge=True))
t_km = timed(lambda: ops.corr_fwd(th, ph, bl, 1e-10, h, w))
part = ops.corr_fwd(th, ph, bl, 1e-10, h, w, defer_merge=True)
res = ops.corr_fwd(th, ph, bl, 1e-10, h, w)
t_mp = timed(lambda: ops.pack_color_input(fr, part, None, last), 50, 100)
t_p = timed(lambda: ops.pack_color_input(fr, res["y_up"], res["sim_up"], last), 50, 100)
print(f"P = {P}: corr_fwd_kernel alone {t_k:.2f} us = {FLOPS / t_k / 1e6 / 157.3:.4f} of the fp32 MFMA peak; + stand-alone merge {t_km:.2f} us = "
      f"{FLOPS / t_km / 1e6 / 157.3:.4f}; consumer: merge + pack in one launch {t_mp:.2f} us, pack alone {t_p:.2f} us")
