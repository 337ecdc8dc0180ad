"""Timing of the Winograd F(2x2,3x3) path against the direct engine on the network's 3x3 layer shapes
(round-robin after a warm-up, min over rounds).  Run on the GPU box:  python tools/conv_wino_probe.py"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deep-exemplar-based-video-colorization_amd"))
import torch
from dvc_amd import ops

SHAPES = [  # Cin, Cout, H, W, dil, in_up
    (512, 512, 27, 48, 1, 1), (512, 512, 27, 48, 2, 1), (256, 256, 54, 96, 1, 1), (128, 128, 108, 192, 1, 1),
    (128, 128, 216, 384, 1, 1), (128, 128, 108, 192, 1, 2), (64, 64, 216, 384, 1, 1), (64, 128, 216, 384, 1, 1),
    (256, 128, 54, 96, 1, 2), (512, 256, 27, 48, 1, 2), (128, 256, 54, 96, 1, 1), (256, 512, 27, 48, 1, 1),
    (64, 128, 108, 192, 1, 1), (512, 512, 13, 24, 1, 1),
]


def timeit(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    out = []
    for (Cin, Cout, H, W, dil, up) in SHAPES:
        x = torch.randn(1, Cin, H, W, generator=g).cuda()
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
        b = torch.zeros(Cout).cuda()
        wp, up_ = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
        cands = {"direct": lambda: ops.conv2d(x, wp, b, dil=dil, pad=dil, in_up=up, act=ops.ACT_RELU)}
        cands["wino auto"] = lambda: ops.conv2d_winograd(x, up_, b, dil=dil, in_up=up, act=ops.ACT_RELU)
        for cfg in range(12):
            if cfg < 4 and Cout % 128:
                continue
            for S in (1, 2, 3, 4, 5, 6, 8):
                def f(cfg=cfg, S=S):
                    return ops.conv2d_winograd(x, up_, b, dil=dil, in_up=up, act=ops.ACT_RELU, cfg=cfg, split_k=S)
                try:
                    f()
                except RuntimeError:
                    continue
                cands[f"wino cfg{cfg} S{S}"] = f
        yd = cands["direct"]()
        yw = cands["wino auto"]()
        err = ((yd - yw).abs().max() / yd.abs().max()).item()
        for _ in range(30):
            cands["direct"]()
        best = {k: float("inf") for k in cands}
        for rnd in range(3):
            for k, f in cands.items():
                best[k] = min(best[k], timeit(f))
        OH, OW = yd.shape[2:]
        gf = 2.0 * Cin * Cout * 9 * OH * OW / 1e9
        top = sorted((v, k) for k, v in best.items() if k.startswith("wino cfg"))[:3]
        line = (f"{Cin:4d}->{Cout:4d} {H:3d}x{W:3d} d{dil} up{up} {gf:6.2f} GF: direct {best['direct']:6.1f} us, "
                f"wino auto {best['wino auto']:6.1f} us ({gf / best['wino auto'] * 1e3:6.1f} TF eff), best "
                + ", ".join(f"{k[5:]} {v:.1f}" for v, k in top) + f"; max |diff| / max |y| = {err:.1e}")
        print(line, flush=True)
        out.append(dict(Cin=Cin, Cout=Cout, H=H, W=W, dil=dil, in_up=up, gflop=gf, us=best))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/conv_wino_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
