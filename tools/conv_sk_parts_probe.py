"""Timing experiment on the GPU box: the stream-K convolution (cfg 9: 64x64 tile, buffer-descriptor staging) with parts
of its unit loop removed (results are wrong, timing only): bit 0 no barrier, bit 1 no staging, bit 2 no LDS operand reads.
Variants are timed round-robin, several rounds, after a long warm-up (the first timings of a process run at a lower clock)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
import torch  # noqa: E402

from dvc_amd import ops  # noqa: E402

dev = torch.device("cuda")
names = {41: "sk9 full", 43: "sk11 (32co x 128px)", 53: "MFMA + adds only", -1: "old engine (auto)", 4: "old cfg4 split 3"}
for (ci, co, H, W) in ((256, 256, 54, 96), (512, 512, 27, 48), (128, 128, 108, 192), (128, 128, 216, 384)):
    x = torch.randn(1, ci, H, W, device=dev)
    wt = torch.randn(ci, 9, co, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    out = torch.empty(1, co, H, W, device=dev)
    flop = 2.0 * ci * co * 9 * H * W
    for per_cu in (1, 2):
        def run(cfg):
            ops.conv2d(x, wt, b, pad=1, act=1, cfg=cfg, split_k=per_cu if cfg >= 32 else (3 if cfg == 4 else 0), out=out)
        for _ in range(200):
            run(41)
        res = {c: [] for c in names}
        for rnd in range(5):
            for c in names:
                run(c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(c)
                e1.record()
                torch.cuda.synchronize()
                res[c].append(e0.elapsed_time(e1) * 100)
        print(f"{ci}->{co} {H}x{W}, {per_cu} workgroup(s)/CU (MFMA floor {flop / 157.3e12 * 1e6:.1f} us): " +
              " | ".join(f"{names[c]} {min(v):.1f}" for c, v in res.items()), flush=True)
