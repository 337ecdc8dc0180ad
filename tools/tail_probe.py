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
