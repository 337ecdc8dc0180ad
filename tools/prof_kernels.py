"""Short kernel-only workload for rocprofv3 --pmc passes: the fused correlation at P=5184 and a few
representative conv layers (each launched a handful of times)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
import torch  # noqa: E402

from dvc_amd import ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
R = int(os.environ.get("DVC_PROF_R", "1"))
if R > 1:
    # the multi-reference pass (ClipColorizer.set_exemplars): ColorVidNet's layer shapes at batch R under the batch-aware launch
    # plan (DVC_CONV_BATCH_PLAN) — what r04's PMC pass reports next to the single-image launches
    with ops.batch_plan(True):
        for (ci, co, H, W, dil, up) in [(256, 256, 54, 96, 1, 1), (512, 512, 27, 48, 1, 1), (512, 512, 27, 48, 2, 1),
                                        (128, 128, 108, 192, 1, 1), (64, 64, 216, 384, 1, 1), (128, 128, 216, 384, 1, 1),
                                        (128, 256, 54, 96, 1, 1), (256, 512, 27, 48, 1, 1)]:
            x = torch.randn(R, ci, H, W, device=dev)
            u = ops.pack_winograd_weight(torch.randn(co, ci, 3, 3, device=dev) * 0.05)
            b = torch.randn(co, device=dev)
            for _ in range(3):
                ops.conv2d_winograd(x, u, b, dil=dil, in_up=up, act=1)
    torch.cuda.synchronize()
    print("done (batch %d, batch plan)" % R)
    sys.exit(0)
h, w = 54, 96
P = h * w
th = ops.corr_prepare(torch.randn(1, 256, P, generator=g).to(dev))
ph = ops.corr_prepare(torch.randn(1, 256, P, generator=g).to(dev))
bl = torch.randn(1, 3, P, generator=g).to(dev)
for _ in range(5):
    ops.corr_fwd(th, ph, bl, 1e-10, h, w)
# (Cin, Cout, H, W, dil, cfg, split_k): the configurations the autotuner picks for these layers
shapes = [(512, 512, 27, 48, 1, 4, 3), (256, 256, 54, 96, 1, 4, 1), (128, 128, 216, 384, 1, 4, 1),
          (128, 128, 108, 192, 1, 4, 1), (512, 512, 27, 48, 2, 4, 3), (64, 64, 216, 384, 1, 3, 1),
          (256, 256, 54, 96, 1, 0, 2),
          # stream-K (cfg 32 + tile configuration, split_k = workgroups per CU)
          (256, 256, 54, 96, 1, 36, 2), (512, 512, 27, 48, 1, 36, 2), (128, 128, 216, 384, 1, 36, 2)]
for (ci, co, H, W, dil, cfg, sk) in shapes:
    x = torch.randn(1, ci, H, W, device=dev)
    wt = torch.randn(ci, 9, co, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    for _ in range(3):
        ops.conv2d(x, wt, b, dil=dil, pad=dil, act=1, cfg=cfg, split_k=sk)
# Winograd F(2x2,3x3) kernel (library's own shape / split choice) on the layer shapes that carry the frame
for (ci, co, H, W, dil, up) in [(256, 256, 54, 96, 1, 1), (512, 512, 27, 48, 1, 1), (512, 512, 27, 48, 2, 1),
                                (128, 128, 108, 192, 1, 1), (64, 64, 216, 384, 1, 1), (128, 128, 216, 384, 1, 1),
                                (128, 128, 108, 192, 1, 2)]:
    x = torch.randn(1, ci, H, W, device=dev)
    u = ops.pack_winograd_weight(torch.randn(co, ci, 3, 3, device=dev) * 0.05)
    b = torch.randn(co, device=dev)
    for _ in range(3):
        ops.conv2d_winograd(x, u, b, dil=dil, in_up=up, act=1)
# r06: the weights-in-registers direct kernel (csrc/conv_ws.hip) on the engine-map layers it takes
for (ci, co, H, W) in [(64, 64, 216, 384), (32, 64, 216, 384), (64, 128, 108, 192), (128, 128, 108, 192), (128, 256, 54, 96), (128, 128, 216, 384)]:
    x = torch.randn(1, ci, H, W, device=dev)
    u = ops.pack_ws_weight(torch.randn(co, ci, 3, 3, device=dev) * 0.05)
    b = torch.randn(co, device=dev)
    for _ in range(3):
        ops.conv2d_ws(x, u, b, co, act=1)
torch.cuda.synchronize()
print("done")
