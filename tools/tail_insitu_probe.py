"""What the per-frame tail costs INSIDE the pipelined driver (ClipColorizer.clip_rgb) as a function of how many frames share one
set of tail launches (`tail_batch`), with and without the WLS filter, next to the bare recurrence (`clip`)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dvc_amd import ops, synth, tail  # noqa: E402
from dvc_amd.frame import ClipColorizer  # noqa: E402

H, W, K = 216, 384, 32
dev = torch.device("cuda")
nets, _ = bench.build_nets(dev)
ops.set_autotune(True)
cc = ClipColorizer(*nets, temperature=1e-10, graph=True)
cc.set_exemplar(tail.downsample_half(synth.synth_lab(synth.EXEMPLAR_SEED, 2 * H, 2 * W).to(dev)))
large = [synth.synth_lab(synth.FRAME_SEED0 + i, 2 * H, 2 * W).to(dev) for i in range(K)]
small = [tail.downsample_half(f) for f in large]
for f in small[:4]:
    cc.frame(f, torch.zeros_like(f))
cc.clip(small[:6])
cc.clip_rgb(large[:8])
torch.cuda.synchronize()


def fps(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return K / sorted(ts)[len(ts) // 2]


for rnd in range(2):
    print(f"round {rnd}: clip {fps(lambda: cc.clip(small)):.1f} frames/s")
    print(f"round {rnd}: downsample x{K} + clip {fps(lambda: cc.clip([tail.downsample_half(f) for f in large])):.1f}")
    for wls in (False, True):
        for tb in (1, 4, 8, 16, 32):
            print(f"round {rnd}: clip_rgb wls={wls} tail_batch={tb}: {fps(lambda: cc.clip_rgb(large, wls_filter_on=wls, tail_batch=tb)):.1f}")
