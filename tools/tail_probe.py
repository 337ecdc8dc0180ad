"""Timing of the clip-driver tail (test.py:98-116) on the GPU at the reference's sizes:
ab 1x2x216x384 -> x2 -> WLS at 432x768 -> 8-bit RGB; and of the oracle (numpy, 1 core) beside it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dvc_amd import tail  # noqa: E402
from oracle import tail_oracle as T  # noqa: E402

H, W = 216, 384
g = torch.Generator().manual_seed(0)
L = torch.rand(1, 1, 2 * H, 2 * W, generator=g) * 100 - 50
lab = torch.cat((L, torch.zeros(1, 2, 2 * H, 2 * W)), 1).cuda()
ab = (torch.randn(1, 2, H, W, generator=g) * 25).cuda()
for _ in range(3):
    tail.frame_tail(lab, ab)
for name, fn in (("whole tail", lambda: tail.frame_tail(lab, ab)),
                 ("no WLS", lambda: tail.frame_tail(lab, ab, wls_filter_on=False))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"GPU {name}: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us per frame")
t0 = time.perf_counter()
T.frame_tail(L.numpy(), ab.cpu().numpy())
print(f"CPU oracle (numpy, 1 core): {(time.perf_counter() - t0) * 1e3:.0f} ms per frame")

# ---- whole device-side loop: full-resolution Lab frames in -> 8-bit RGB out (ClipColorizer.clip_rgb)
import bench  # noqa: E402
from dvc_amd import ops, synth  # noqa: E402
from dvc_amd.frame import ClipColorizer  # noqa: E402

dev = torch.device("cuda")
nets, _ = bench.build_nets(dev)
ops.set_autotune(True)
cc = ClipColorizer(*nets, temperature=1e-10)
cc.set_exemplar(tail.downsample_half(synth.synth_lab(synth.EXEMPLAR_SEED, 2 * H, 2 * W).to(dev)))
K = 30
large = [synth.synth_lab(synth.FRAME_SEED0 + i, 2 * H, 2 * W).to(dev) for i in range(K)]
small = [tail.downsample_half(f) for f in large]
for f in small[:4]:
    cc.frame(f, torch.zeros_like(f))          # autotune, sequentially
cc.clip(small[:4])
cc.clip_rgb(large[:4])
torch.cuda.synchronize()
for name, fn in (("clip (network resolution in, ab out)", lambda: cc.clip(small)),
                 ("clip_rgb (432x768 Lab in, RGB8 out, WLS on)", lambda: cc.clip_rgb(large)),
                 ("clip_rgb, WLS off", lambda: cc.clip_rgb(large, wls_filter_on=False))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    print(f"{name}: {K / (time.perf_counter() - t0):.1f} frames/s")

# ---- ingest: 1080p 8-bit RGB -> CenterPad(432x768) -> centred Lab
import numpy as np  # noqa: E402

rgb = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)).to(dev)
for _ in range(3):
    tail.frame_ingest(rgb, (2 * H, 2 * W))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    tail.frame_ingest(rgb, (2 * H, 2 * W))
e1.record()
torch.cuda.synchronize()
print(f"GPU ingest 1080x1920 RGB8 -> 432x768 Lab: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us per frame")
frames8 = [rgb.roll(7 * i, 1) for i in range(K)]
cc.colorize_video(frames8[:4], frames8[0].flip(0).contiguous(), image_size=(2 * H, 2 * W))
torch.cuda.synchronize()
t0 = time.perf_counter()
cc.colorize_video(frames8, frames8[0].flip(0).contiguous(), image_size=(2 * H, 2 * W))
torch.cuda.synchronize()
print(f"colorize_video (1080p RGB8 in -> 432x768 RGB8 out, exemplar prep included): {K / (time.perf_counter() - t0):.1f} frames/s")
