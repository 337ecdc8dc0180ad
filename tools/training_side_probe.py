"""Timing of the training-side reuse at the sizes the training caller runs
(/root/reference/train.py:44,402-427,649-668: 216x384 crops, batch 16, temperature 0.01):
  * the differentiable fused correlation (dvc_amd.corr_autograd): forward, forward + backward at 54 x 96, B = 16;
  * ContextualLoss_forward / ContextualLoss (dvc_amd.contextual) forward + backward on relu5_1 (512 x 13 x 24),
    relu4_1 (512 x 27 x 48) and the x0.5 relu3_1 (256 x 27 x 48), B = 16;
next to the oracle's CPU autograd (oracle.correlate / oracle.contextual_oracle, fp32, torch CPU) on B = 2 of the same
inputs in the same run.  GPU box: python tools/training_side_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from dvc_amd import ops  # noqa: E402
from dvc_amd.contextual import ContextualLoss, ContextualLoss_forward  # noqa: E402
from dvc_amd.corr_autograd import fused_correlation  # noqa: E402
from oracle import contextual_oracle as CO  # noqa: E402
from oracle import dvc_oracle as O  # noqa: E402

dev = torch.device("cuda")
try:
    avail = len(os.sched_getaffinity(0))
except AttributeError:
    avail = os.cpu_count() or 1
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        avail = max(1, min(avail, int(int(q) / int(p))))
except (OSError, ValueError):
    pass
torch.set_num_threads(max(1, min(32, avail)))
ops.set_autotune(True)
PEAK = 157.3


def gpu_time(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


_filler = None


def device_time(fn, reps):
    """GPU time of `fn` with the launch queue primed: ~12 ms of filler GEMMs are enqueued first, so the host issues all of fn's
    launches while the GPU is still busy and the events bracket back-to-back kernel execution only — what a training step sees
    (its launches are queued behind other work), as opposed to gpu_time's wall clock of a call on an idle GPU, which for the
    small maps is the host's ~25 Python / ctypes calls per loss."""
    global _filler
    if _filler is None:
        _filler = (torch.randn(8192, 8192, device=dev), torch.randn(8192, 8192, device=dev), torch.empty(8192, 8192, device=dev))
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2 + reps):
        torch.mm(_filler[0], _filler[1], out=_filler[2])
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def cpu_time(fn, reps=2):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def unit(t):
    t = t - t.mean(-1, keepdim=True)
    return t / t.norm(dim=1, keepdim=True)


print(f"host threads for the CPU legs: {torch.get_num_threads()}")
# ---------------------------------------------------------------------------------- fused correlation, 54 x 96
h, w, C, T = 54, 96, 256, 0.01
P = h * w
g = torch.Generator().manual_seed(0)
for B in (2, 16):
    th = unit(torch.randn(B, C, P, generator=g)).to(dev)
    ph = unit(torch.randn(B, C, P, generator=g)).to(dev)
    lab = (torch.randn(B, 3, 4 * h, 4 * w, generator=g) * 30).to(dev)
    blab = ops.avgpool4x4(lab).view(B, 3, P)
    gy, gs = torch.randn(B, 3, h, w, generator=g).to(dev), torch.randn(B, 1, h, w, generator=g).to(dev)

    def fwd():
        with torch.no_grad():
            return fused_correlation(th, ph, blab, T, h, w)

    def fwd_bwd():
        a, b_ = th.detach().requires_grad_(True), ph.detach().requires_grad_(True)
        y, sim, _ = fused_correlation(a, b_, blab, T, h, w)
        ((y * gy).sum() + (sim * gs).sum()).backward()

    t_f, t_fb = gpu_time(fwd, 5), gpu_time(fwd_bwd, 3)
    fl_f, fl_b = B * 2.0 * P * P * (C + 3), B * 6.0 * P * P * C        # backward: F recompute + d phi + d theta GEMMs
    print(f"fused_correlation {h}x{w} B={B} T={T}: forward {t_f:.2f} ms ({fl_f / t_f / 1e9:.1f} TFLOP/s = {fl_f / t_f / 1e9 / PEAK:.2f} of the fp32 "
          f"matrix peak), forward + backward {t_fb:.2f} ms (backward {t_fb - t_f:.2f} ms: {fl_b / (t_fb - t_f) / 1e9:.1f} TFLOP/s on its "
          f"three recompute GEMMs = {fl_b / (t_fb - t_f) / 1e9 / PEAK:.2f})", flush=True)
    if B == 2:
        thc, phc, labc, gyc, gsc = th.cpu(), ph.cpu(), lab.cpu(), gy.cpu(), gs.cpu()

        def cpu_fb():
            a, b_ = thc.clone().requires_grad_(True), phc.clone().requires_grad_(True)
            y, sim, _ = O.correlate(a, b_, labc, T)
            ((y * gyc).sum() + (sim * gsc).sum()).backward()

        t_c = cpu_time(cpu_fb, 1)
        print(f"    oracle (torch CPU fp32 autograd through NonlocalNet.py:477-500, materialised P x P), B=2: forward + backward "
              f"{t_c:.0f} ms  -> HIP {t_c / t_fb:.0f}x", flush=True)

# ---------------------------------------------------------------------------------- contextual losses
for (name, Cc, hh, ww) in (("relu5_1", 512, 13, 24), ("relu4_1", 512, 27, 48), ("relu3_1 x0.5", 256, 27, 48)):
    for cls, oracle_fn, label in ((ContextualLoss_forward, CO.contextual_loss_forward, "ContextualLoss_forward"),
                                  (ContextualLoss, CO.contextual_loss, "ContextualLoss")):
        res = {}
        for B in (2, 16):
            gg = torch.Generator().manual_seed(7)
            X = torch.relu(torch.randn(B, Cc, hh, ww, generator=gg)).to(dev)
            Y = torch.relu(torch.randn(B, Cc, hh, ww, generator=gg)).to(dev)
            mod = cls()

            def fb():
                x = X.detach().requires_grad_(True)
                mod(x, Y).mean().backward()

            res[B] = gpu_time(fb, 3)
            if B == 16:
                res["dev"] = device_time(fb, 4)
            if B == 2:
                Xc, Yc = X.cpu(), Y.cpu()

                def cpu_fb():
                    x = Xc.clone().requires_grad_(True)
                    oracle_fn(x, Yc).mean().backward()

                res["cpu"] = cpu_time(cpu_fb, 1)
        N = hh * ww
        fl = 16 * 2.0 * N * N * Cc * 3          # S forward, S recompute, d Xn
        print(f"{label:24s} {name:13s} ({Cc} x {hh} x {ww}): forward + backward B=16 {res[16]:.2f} ms ({fl / res[16] / 1e9:.1f} TFLOP/s on "
              f"its three N x N GEMMs; with the launch queue primed {res['dev']:.2f} ms = {fl / res['dev'] / 1e9:.1f} TFLOP/s), B=2 {res[2]:.2f} ms; "
              f"oracle CPU autograd B=2 {res['cpu']:.0f} ms -> HIP {res['cpu'] / res[2]:.0f}x",
              flush=True)
