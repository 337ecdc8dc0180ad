"""Timing experiment: how much of a DMA-staged conv layer's time is exposed global->LDS latency?
Runs a few layers with the staging DMA partly disabled (results are wrong; timing only)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
os.environ.setdefault("DVC_DEBUG_LIB", "1")   # the dvc_debug_* hooks live in the -DDVC_DEBUG build (make -C csrc DEBUG=1)
import torch  # noqa: E402

from dvc_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.dvc_debug_conv_variant.restype = None
lib.dvc_debug_conv_variant.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
shapes = [(128, 128, 216, 384, 1, 3, 1), (128, 128, 216, 384, 1, 4, 1), (256, 256, 54, 96, 1, 4, 1),
          (512, 512, 27, 48, 1, 4, 1), (512, 512, 27, 48, 1, 4, 3), (64, 64, 216, 384, 1, 4, 1)]
names = {0: "normal", 1: "no DMA after chunk 0", 2: "no patch DMA", 3: "no weight DMA"}
for (ci, co, H, W, dil, cfg, sk) in shapes:
    x = torch.randn(1, ci, H, W, device=dev)
    wt = torch.randn(ci, 9, co, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    out = torch.empty(1, co, H, W, device=dev)
    row = []
    for v in (0, 1, 2, 3):
        lib.dvc_debug_conv_variant(v)
        for _ in range(3):
            ops.conv2d(x, wt, b, dil=dil, pad=dil, act=1, cfg=cfg, split_k=sk, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d(x, wt, b, dil=dil, pad=dil, act=1, cfg=cfg, split_k=sk, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        row.append(f"{names[v]} {us:.0f}us ({2e-6 * ci * co * 9 * H * W / us:.0f} TF)")
    lib.dvc_debug_conv_variant(0)
    print(f"{ci}->{co} {H}x{W} cfg{cfg} sk{sk}: " + " | ".join(row))
