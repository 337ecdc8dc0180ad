"""A/B of Winograd-kernel variants on the layers that carry the frame's convolution time (GPU box, -DDVC_DEBUG build:
`make -C deep-exemplar-based-video-colorization_amd/csrc DEBUG=1`).  Variants are selected with dvc_debug_conv_variant:
  0   production (packed transform arithmetic, v_pk_add_f32)
  16  scalar transform arithmetic (v_add_f32 / v_sub_f32): MI355X_MICROARCH.md prices a packed fp32 VALU beside MFMAs
      above the scalar pair
Round-robin over the variants after a clock warm-up, minimum over rounds; the reduce launch of split layers is included.
Results are checked equal between variants (same arithmetic, same order: bit-identical)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
os.environ.setdefault("DVC_DEBUG_LIB", "1")
import torch  # noqa: E402

from dvc_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda")
lib = _lib.load()
VARIANTS = [int(v) for v in os.environ.get("WINO_VARIANTS", "0,16").split(",")]
# (count per frame, Cin, Cout, H, W, dil, in_up)
LAYERS = [(13, 256, 256, 54, 96, 1, 1), (8, 512, 512, 27, 48, 1, 1), (6, 512, 512, 27, 48, 2, 1), (5, 128, 128, 108, 192, 1, 1),
          (2, 64, 64, 216, 384, 1, 1), (1, 128, 128, 216, 384, 1, 1), (1, 64, 128, 216, 384, 1, 1), (2, 128, 256, 54, 96, 1, 1),
          (2, 256, 512, 27, 48, 1, 1), (2, 64, 128, 108, 192, 1, 1), (1, 128, 128, 108, 192, 1, 2), (1, 256, 128, 54, 96, 1, 2),
          (1, 512, 256, 27, 48, 1, 2), (2, 512, 512, 13, 24, 1, 1)]


def timeit(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {v: 0.0 for v in VARIANTS}
CFGS = [tuple(int(t) for t in p.split(":")) for p in os.environ.get("WINO_CFGS", "").split(",") if p]
tot_forced = [0.0]
for (cnt, ci, co, H, W, dil, up) in LAYERS:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, ci, H, W, generator=g).to(dev)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    u = ops.pack_winograd_weight(w)
    run = lambda: ops.conv2d_winograd(x, u, b, dil=dil, in_up=up, act=ops.ACT_RELU)   # noqa: E731
    outs, best = {}, {v: 1e9 for v in VARIANTS}
    for v in VARIANTS:
        lib.dvc_debug_conv_variant(v)
        outs[v] = run().clone()
    for _ in range(30):
        run()
    for rnd in range(5):
        for v in VARIANTS:
            lib.dvc_debug_conv_variant(v)
            best[v] = min(best[v], timeit(run))
    lib.dvc_debug_conv_variant(0)
    same = all(torch.equal(outs[v], outs[VARIANTS[0]]) for v in VARIANTS)
    gf = 2.0 * ci * co * 9 * (H * up) * (W * up) / 1e9
    line = (f"x{cnt:<2d} {ci:4d}->{co:4d} {H:3d}x{W:3d} d{dil} up{up} {gf:6.2f} GF: " +
            "  ".join(f"v{v}: {best[v]:6.1f} us" for v in VARIANTS) + f"  identical: {same}")
    # forced workgroup shapes / splits (WINO_CFGS="cfg:split,..."; cfg = 4 * shape + tile-block index): best of them per layer
    if CFGS:
        cb = {}
        for (cfg, sk) in CFGS:
            try:
                f = lambda: ops.conv2d_winograd(x, u, b, dil=dil, in_up=up, act=ops.ACT_RELU, cfg=cfg, split_k=sk)   # noqa: E731
                ref = f()
                assert (ref - outs[VARIANTS[0]]).abs().max().item() <= 2e-5 * outs[VARIANTS[0]].abs().max().item() + 1e-6
                cb[(cfg, sk)] = min(timeit(f) for _ in range(3))
            except RuntimeError:
                pass
        if cb:
            k = min(cb, key=cb.get)
            line += f"  | forced best cfg {k[0]} split {k[1]}: {cb[k]:6.1f} us"
            tot_forced[0] += cnt * min(cb[k], best[VARIANTS[0]])
    print(line, flush=True)
    for v in VARIANTS:
        tot[v] += cnt * best[v]
print("per frame (these layers): " + "  ".join(f"v{v}: {tot[v] / 1e3:.3f} ms" for v in VARIANTS) +
      (f"  | with the forced shapes where they win: {tot_forced[0] / 1e3:.3f} ms" if CFGS else ""))
