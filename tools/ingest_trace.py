"""Kernel-only workload for `rocprofv3 --kernel-trace --stats`: the frame ingest of ONE 1080x1920 RGB8 frame, 30 times
(dvc_amd.tail.frame_ingest: CenterPad's anti-aliased resize to 432x768, RGB -> Lab) — which kernels the ~118 us per frame of
tools/tail_probe.py consist of."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
import torch  # noqa: E402

from dvc_amd import tail  # noqa: E402

g = torch.Generator().manual_seed(0)
rgb = torch.randint(0, 256, (1080, 1920, 3), generator=g, dtype=torch.uint8).cuda()
for _ in range(30):
    tail.frame_ingest(rgb, (432, 768))
torch.cuda.synchronize()
print("done")
