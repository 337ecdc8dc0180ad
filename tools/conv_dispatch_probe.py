"""Timing experiment: where do the workgroups of a conv layer run, and for how long?
Per workgroup: s_memtime at entry / after the chunk loop, HW_ID and XCC_ID."""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
os.environ.setdefault("DVC_DEBUG_LIB", "1")   # the dvc_debug_* hooks live in the -DDVC_DEBUG build (make -C csrc DEBUG=1)
import torch  # noqa: E402

from dvc_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.dvc_debug_conv_trace.restype = None
lib.dvc_debug_conv_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
shapes = [(256, 256, 54, 96, 1, 4, 1), (256, 256, 54, 96, 1, 4, 2), (256, 256, 54, 96, 1, 4, 3),
          (256, 256, 54, 96, 1, 3, 1), (256, 256, 54, 96, 1, 3, 2), (256, 256, 54, 96, 1, 3, 3), (256, 256, 54, 96, 1, 3, 4),
          (512, 512, 27, 48, 1, 4, 3), (128, 128, 216, 384, 1, 4, 1), (128, 128, 108, 192, 1, 4, 1)]
for (ci, co, H, W, dil, cfg, sk) in shapes:
    x = torch.randn(1, ci, H, W, device=dev)
    wt = torch.randn(ci, 9, co, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    out = torch.empty(1, co, H, W, device=dev)
    for _ in range(3):
        ops.conv2d(x, wt, b, dil=dil, pad=dil, act=1, cfg=cfg, split_k=sk, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv2d(x, wt, b, dil=dil, pad=dil, act=1, cfg=cfg, split_k=sk, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    buf = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
    lib.dvc_debug_conv_trace(ctypes.c_void_p(buf.data_ptr()))
    ops.conv2d(x, wt, b, dil=dil, pad=dil, act=1, cfg=cfg, split_k=sk, out=out)
    torch.cuda.synchronize()
    lib.dvc_debug_conv_trace(None)
    t = buf.view(-1, 4).cpu()
    t = t[t[:, 0] != 0]
    nwg = t.shape[0]
    hw, xcc = t[:, 2], t[:, 3] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_cu = collections.Counter(key.tolist())
    hist = collections.Counter(per_cu.values())
    dur = (t[:, 1] - t[:, 0]).double()
    # per-CU span: first entry to last loop end (same clock within a CU)
    spans = []
    for k in per_cu:
        m = key == k
        spans.append((t[m, 1].max() - t[m, 0].min()).item())
    spans = torch.tensor(spans).double()
    # concurrency: for each CU, order of WG start times
    late = 0
    for k in per_cu:
        m = key == k
        st = t[m, 0].sort().values
        en = t[m, 1].sort().values
        if len(st) > 1 and st[-1] > en[0]:
            late += 1
    print(f"{ci}->{co} {H}x{W} cfg{cfg} sk{sk}: {us:.0f} us, {nwg} WGs on {len(per_cu)} CUs, WGs/CU histogram {dict(sorted(hist.items()))}; "
          f"WG loop ticks mean {dur.mean():.0f} min {dur.min():.0f} max {dur.max():.0f}; per-CU span ticks mean {spans.mean():.0f} max {spans.max():.0f}; "
          f"CUs that started a WG after another of theirs finished: {late}")
    # duration by number of WGs sharing the CU
    for n in sorted(hist):
        ks = [k for k, v in per_cu.items() if v == n]
        m = torch.isin(key, torch.tensor(ks))
        print(f"   CUs with {n} WGs: mean WG loop ticks {dur[m].mean():.0f}")
