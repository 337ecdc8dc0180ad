"""A/B on the GPU box: fused correlation with one accumulator chain (r01 kernel, debug switch) vs two (default),
P = 5184 and 20736, T = 1e-10 and 0.01; and both with the softmax elided (timing experiment)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
import torch  # noqa: E402

from dvc_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda")
lib = _lib.load()
for (h, w) in ((54, 96), (108, 192)):
    P = h * w
    g = torch.Generator().manual_seed(1)
    th = ops.corr_prepare(torch.randn(1, 256, P, generator=g).to(dev))
    ph = ops.corr_prepare(torch.randn(1, 256, P, generator=g).to(dev))
    bl = torch.randn(1, 3, P, generator=g).to(dev)
    flops = 2.0 * P * P * 259
    outs = {}
    for name, variant in (("single", 0), ("dual", 2), ("nosoftmax", 1), ("antiphase", 3)):
        lib.dvc_debug_corr_variant(variant)
        for T in ((1e-10, 0.01) if variant != 1 else (1e-10,)):
            for _ in range(3):
                r = ops.corr_fwd(th, ph, bl, T, h, w, want_small=True, want_argmax=True)
            reps = 20 if P < 10000 else 6
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.corr_fwd(th, ph, bl, T, h, w)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            outs[(name, T)] = r
            print(f"P={P} {name:15s} T={T:g}: {us:7.1f} us  {flops / us / 1e6:6.1f} TFLOP/s  {flops / us / 1e6 / 157.3 * 100:4.1f} % of fp32 MFMA peak", flush=True)
    lib.dvc_debug_corr_variant(0)
    for T in (1e-10, 0.01):
        a, b = outs[("antiphase", T)], outs[("single", T)]
        print(f"P={P} T={T:g}: antiphase vs single  argmax differs on {(a['argmax'] != b['argmax']).sum().item()} rows, "
              f"sim max diff {(a['sim_small'] - b['sim_small']).abs().max().item():.2e}, y max diff {(a['y_small'] - b['y_small']).abs().max().item():.2e}")
