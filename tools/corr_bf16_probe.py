"""Timing of the bf16 candidate-filter correlation (configs[4]) next to the fp32 kernel, P = 5184 and 20736."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
import torch  # noqa: E402

from dvc_amd import ops  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, n=20):
    for _ in range(60):                   # warm clock (the first milliseconds after a load change run slower)
        fn()
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for h, w in ((54, 96), (108, 192)):
    P = h * w
    g = torch.Generator().manual_seed(1)
    tr = torch.randn(1, 256, P, generator=g).to(dev)
    pr = torch.randn(1, 256, P, generator=g).to(dev)
    bl = torch.randn(1, 3, P, generator=g).to(dev)
    th, ph = ops.corr_prepare(tr), ops.corr_prepare(pr)
    tb, pb = ops.corr_prepare_bf16(tr), ops.corr_prepare_bf16(pr)
    t32 = timeit(lambda: ops.corr_fwd(th, ph, bl, 1e-10, h, w))
    t16 = timeit(lambda: ops.corr_fwd_bf16(tb, pb, bl, 1e-10, h, w))
    tp32 = timeit(lambda: ops.corr_prepare(tr))
    tp16 = timeit(lambda: ops.corr_prepare_bf16(tr))
    a = ops.corr_fwd(th, ph, bl, 1e-10, h, w, want_argmax=True)
    b = ops.corr_fwd_bf16(tb, pb, bl, 1e-10, h, w, want_argmax=True)
    same = (a["argmax"] == b["argmax"]).float().mean().item()
    print(f"P={P}: fp32 {t32:.0f} us, bf16 filter + fp32 re-score {t16:.0f} us  (prepare: {tp32:.0f} / {tp16:.0f} us); "
          f"argmax agreement {same * 100:.2f} %, max |y diff| {(a['y_up'] - b['y_up']).abs().max().item():.1e}")
