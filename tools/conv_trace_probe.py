"""For rocprofv3 --kernel-trace: a few conv layers under a few configurations, back to back, so that the trace shows the
duration of the main kernel, of the fixup / split-K reduce kernel, and the gaps between them."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
import torch  # noqa: E402

from dvc_amd import ops  # noqa: E402

dev = torch.device("cuda")
for (ci, co, H, W) in ((256, 256, 54, 96), (512, 512, 27, 48)):
    x = torch.randn(1, ci, H, W, device=dev)
    wt = torch.randn(ci, 9, co, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    out = torch.empty(1, co, H, W, device=dev)
    for cfg, sk in ((36, 2), (36, 1), (4, 3), (4, 1)):
        for _ in range(30):
            ops.conv2d(x, wt, b, pad=1, act=1, cfg=cfg, split_k=sk, out=out)
        torch.cuda.synchronize()
