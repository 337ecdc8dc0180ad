"""A/B on the GPU box: variants of the fused correlation kernel (debug switch), P = 5184 and 20736, timed ROUND-ROBIN
over several rounds after a long warm-up (the first timings of a process run at a lower clock; min over rounds)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
os.environ.setdefault("DVC_DEBUG_LIB", "1")   # the dvc_debug_* hooks live in the -DDVC_DEBUG build (make -C csrc DEBUG=1)
import torch  # noqa: E402

from dvc_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda")
lib = _lib.load()
VARIANTS = (("default", 0), ("no softmax", 1))
for (h, w) in ((54, 96), (108, 192)):
    P = h * w
    g = torch.Generator().manual_seed(1)
    th = ops.corr_prepare(torch.randn(1, 256, P, generator=g).to(dev))
    ph = ops.corr_prepare(torch.randn(1, 256, P, generator=g).to(dev))
    bl = torch.randn(1, 3, P, generator=g).to(dev)
    flops = 2.0 * P * P * 259
    reps = 20 if P < 10000 else 5
    for T in (1e-10, 0.01):
        for _ in range(100 if P < 10000 else 10):
            ops.corr_fwd(th, ph, bl, T, h, w)
        res = {n: [] for n, _ in VARIANTS}
        for rnd in range(5):
            for name, variant in VARIANTS:
                lib.dvc_debug_corr_variant(variant)
                ops.corr_fwd(th, ph, bl, T, h, w)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.corr_fwd(th, ph, bl, T, h, w)
                e1.record()
                torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) / reps * 1e3)
        lib.dvc_debug_corr_variant(0)
        print(f"P={P} T={T:g}: " + " | ".join(f"{n} {min(v):.1f} us ({flops / min(v) / 1e6 / 157.3 * 100:.1f} %)" for n, v in res.items()), flush=True)
