"""ColorVidNet's decoder-block pairs conv(up(a)) + conv_short(b): one dual launch against the two launches (second one with the
first one's output as residual).  GPU box: python tools/conv_dual_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
import torch
from dvc_amd import ops
dev = torch.device("cuda")


def timeit(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0, 0.0]
for (CA, CB, Cout, H, W) in ((512, 256, 256, 54, 96), (256, 128, 128, 108, 192), (128, 64, 128, 216, 384)):
    g = torch.Generator().manual_seed(1)
    xA = torch.randn(1, CA, H // 2, W // 2, generator=g).to(dev)
    xB = torch.randn(1, CB, H, W, generator=g).to(dev)
    uA = ops.pack_winograd_weight((torch.randn(Cout, CA, 3, 3, generator=g) / (CA * 9) ** 0.5).to(dev))
    uB = ops.pack_winograd_weight((torch.randn(Cout, CB, 3, 3, generator=g) / (CB * 9) ** 0.5).to(dev))
    bA, bB = torch.randn(Cout, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    u, b = torch.cat((uA, uB), 1).contiguous(), bA + bB
    dual = lambda: ops.conv2d_winograd_dual(xA, xB, u, b, in_upA=2, act=1)                                   # noqa: E731
    two = lambda: ops.conv2d_winograd(xA, uA, bA, in_up=2, act=1, residual=ops.conv2d_winograd(xB, uB, bB))  # noqa: E731
    for _ in range(30):
        dual(); two()
    td = min(timeit(dual) for _ in range(4))
    tt = min(timeit(two) for _ in range(4))
    tot[0] += td; tot[1] += tt
    print(f"{CA}+{CB}->{Cout} at {H}x{W}: dual {td:.1f} us, two launches {tt:.1f} us", flush=True)
print(f"per frame: dual {tot[0]:.0f} us, two launches {tot[1]:.0f} us")
