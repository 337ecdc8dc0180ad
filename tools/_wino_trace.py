import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deep-exemplar-based-video-colorization_amd"))
import torch
from dvc_amd import ops, _lib
lib = _lib.load()
lib.dvc_debug_conv_trace.argtypes = [ctypes.c_void_p]
g = torch.Generator().manual_seed(0)
for (Cin, Cout, H, W, dil) in ((256, 256, 54, 96, 1), (512, 512, 27, 48, 1), (128, 128, 216, 384, 1)):
    x = torch.randn(1, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
    u = ops.pack_winograd_weight(w)
    for _ in range(30): ops.conv2d_winograd(x, u, None, dil=dil)
    buf = torch.zeros(4096, 8, dtype=torch.int64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.dvc_debug_conv_trace(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(200): ops.conv2d_winograd(x, u, None, dil=dil)
    e0.record(); ops.conv2d_winograd(x, u, None, dil=dil); e1.record(); torch.cuda.synchronize()
    lib.dvc_debug_conv_trace(None)
    b = buf.cpu().double()
    b = b[b[:, 0] > 0]
    t0 = b[:, 0].min()
    us = lambda v: v / 100.0      # s_memrealtime: 100 MHz
    d = [(b[:, k] - b[:, k - 1]) for k in range(1, 6)]
    print(f"{Cin}->{Cout} {H}x{W}: {len(b)} workgroups, launch {e0.elapsed_time(e1)*1e3:.1f} us (with reduce); first entry -> last exit {us(b[:,5].max()-t0):.1f} us; "
          f"entry spread {us(b[:,0].max()-t0):.1f} us")
    for nm, v in zip(("plan", "first DMA + barrier", "K loop", "flush + exchange", "combine + store"), d):
        print(f"    {nm:22s} mean {us(v.mean()):6.2f} us   min {us(v.min()):6.2f}   max {us(v.max()):6.2f}")
    print(f"    exit spread: first exit {us(b[:,5].min()-t0):.1f} us, last exit {us(b[:,5].max()-t0):.1f} us")
