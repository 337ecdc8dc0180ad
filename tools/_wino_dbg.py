import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deep-exemplar-based-video-colorization_amd"))
import torch
from dvc_amd import ops, _lib
lib = _lib.load()
g = torch.Generator().manual_seed(0)
def timeit(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (Cin, Cout, H, W, cfg, S) in ((128, 128, 216, 384, 0, 1), (512, 512, 27, 96, 0, 1)):
    x = torch.randn(1, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
    u = ops.pack_winograd_weight(w)
    f = lambda: ops.conv2d_winograd(x, u, None, cfg=cfg, split_k=S)
    for _ in range(50): f()
    res = {}
    for rnd in range(3):
        for v in (0, 3, 4, 7, 8, 11, 12, 15, 16, 19, 28, 31):
            lib.dvc_debug_conv_variant(v)
            res[v] = min(res.get(v, 1e9), timeit(f))
    lib.dvc_debug_conv_variant(0)
    print(f"{Cin}->{Cout} {H}x{W} cfg{cfg} S{S}: " + ", ".join(f"v{k}: {v:.1f}" for k, v in res.items()), flush=True)
