"""Kernel-only workload for `rocprofv3 --kernel-trace --stats`: the clip-driver tail of ONE 432x768 frame, 30 times
(dvc_amd.tail.frame_tail: x2 bilinear, luminance guide, fast global smoother = WLS, Lab -> RGB8) — which kernels the
219 us per frame of tools/tail_probe.py consist of."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
import torch  # noqa: E402

from dvc_amd import tail  # noqa: E402

H, W = 216, 384
g = torch.Generator().manual_seed(0)
L = torch.rand(1, 1, 2 * H, 2 * W, generator=g) * 100 - 50
lab = torch.cat((L, torch.zeros(1, 2, 2 * H, 2 * W)), 1).cuda()
ab = (torch.randn(1, 2, H, W, generator=g) * 25).cuda()
for _ in range(30):
    tail.frame_tail(lab, ab)
torch.cuda.synchronize()
print("done")
