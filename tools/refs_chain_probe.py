#!/usr/bin/env python
"""Where the multi-reference pass (ClipColorizer.set_exemplars) spends its time: the ColorVidNet chain at batch 1 and at
batch R with the per-image plan (bit-identical to R calls) and with the batch-aware plan (DVC_CONV_BATCH_PLAN), the front end
of one frame, and R correlations — per call, HIP events, one stream, warm clock."""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402

from dvc_amd import ops, synth  # noqa: E402
from dvc_amd.frame import ClipColorizer, warp_color  # noqa: E402
from models.ColorVidNet import ColorVidNet  # noqa: E402
from models.NonlocalNet import VGG19_pytorch, WarpNet  # noqa: E402

H, W = 216, 384
dev = torch.device("cuda")
with contextlib.redirect_stdout(io.StringIO()):
    nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
for m, sd in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
    m.load_state_dict(sd)
    m.eval().to(dev)
vgg, warp, col = nets


def timed(fn, warm=10, reps=30):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator().manual_seed(0)
for R in (1, 2, 4, 6):
    cin = (torch.randn(R, 7, H, W, generator=g) * 20).to(dev)
    t_img = timed(lambda: col(cin))

    def planned():
        with ops.batch_plan(True):
            return col(cin)
    t_plan = timed(planned) if R > 1 else t_img
    rec = ops.conv_record = []
    planned() if R > 1 else col(cin)
    ops.conv_record = None
    print(f"ColorVidNet chain at batch {R}: per-image plan {t_img:.3f} ms ({t_img / R:.3f} per image), batch-aware plan {t_plan:.3f} ms "
          f"({t_plan / R:.3f} per image); {len(rec)} convolution launches")
cc = ClipColorizer(vgg, warp, col, temperature=1e-10)
fr = synth.synth_lab(synth.FRAME_SEED0, H, W).to(dev)
for R in (1, 4):
    cc.set_exemplars([synth.synth_lab(s, H, W).to(dev) for s in (2, 3, 5, 11)[:R]])
    t_front = timed(lambda: warp_color(fr[:, 0:1], cc.IB_lab, None, vgg, warp, col, 0, temperature=1e-10, exemplar_cache=cc.ex_cache,
                                       defer_merge=True))
    print(f"front end of one frame against {R} reference(s) (VGG19 + WarpNet + {R} correlation(s)): {t_front:.3f} ms")
