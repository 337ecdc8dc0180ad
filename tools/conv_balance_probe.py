"""Experiment: throughput of the conv kernel on shapes whose workgroup count exactly fills the machine
(512 = 256 CUs x 2 resident workgroups) vs. awkward counts — how much does tail quantisation cost?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
import torch  # noqa: E402

from dvc_amd import ops  # noqa: E402

dev = torch.device("cuda")


def run(ci, co, H, W, cfg, sk, reps=10):
    x = torch.randn(1, ci, H, W, device=dev)
    wt = torch.randn(ci, 9, co, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    out = torch.empty(1, co, H, W, device=dev)
    for _ in range(3):
        ops.conv2d(x, wt, b, act=1, cfg=cfg, split_k=sk, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d(x, wt, b, act=1, cfg=cfg, split_k=sk, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    return us, 2.0 * co * H * W * ci * 9 / us / 1e6


# cfg3: 32 co x (4 rows x 32 cols); cfg1: 32 co x (8 x 32); cfg2: 64 co x (4 x 32); cfg0: 64 co x (8 x 32)
cases = [
    ("cfg3 512 wg", 256, 256, 64, 128, 3, 1),      # 16*4 px tiles * 8 = 512
    ("cfg3 1024 wg", 256, 256, 128, 128, 3, 1),
    ("cfg3 1536 wg", 256, 256, 192, 128, 3, 1),
    ("cfg3 672 wg", 256, 256, 84, 128, 3, 1),       # 21*4*8
    ("cfg3 768 wg", 256, 256, 96, 128, 3, 1),
    ("cfg1 512 wg", 256, 256, 128, 128, 1, 1),      # 16*4*8
    ("cfg2 512 wg", 256, 256, 128, 128, 2, 1),      # 32*4*4
    ("cfg0 512 wg", 256, 256, 256, 128, 0, 1),      # 32*4*4
    ("cfg0 256 wg", 256, 256, 128, 128, 0, 1),
    ("cfg3 512 wg Cin512", 512, 256, 64, 128, 3, 1),
    ("cfg3 256wg x sk2", 256, 256, 32, 128, 3, 2),
    ("cfg4 512 wg", 256, 256, 32, 128, 4, 1),       # 64co x 2 rows: 16*4*4=256 -> H=64: 512
]
for name, ci, co, H, W, cfg, sk in cases:
    us, tf = run(ci, co, H, W, cfg, sk)
    print(f"{name:22s} Cin={ci} Cout={co} {H}x{W}: {us:7.1f} us  {tf:6.1f} TF/s")
