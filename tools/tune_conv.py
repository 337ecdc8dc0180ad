"""Time every distinct conv2d launch of one 216x384 frame under each tile configuration (GPU box).
Writes gpurun_out/tune_conv.json: per layer shape, microseconds per cfg and the auto choice."""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dvc_amd import ops, synth  # noqa: E402
from dvc_amd.frame import ClipColorizer  # noqa: E402
from models.ColorVidNet import ColorVidNet  # noqa: E402
from models.NonlocalNet import VGG19_pytorch, WarpNet  # noqa: E402

H, W = int(os.environ.get("TUNE_H", 216)), int(os.environ.get("TUNE_W", 384))
dev = torch.device("cuda")
with contextlib.redirect_stdout(io.StringIO()):
    nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
for m, s in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
    m.load_state_dict(s)
    m.eval().to(dev)
cc = ClipColorizer(*nets)
cc.set_exemplar(synth.synth_lab(2, H, W).to(dev))
fr = synth.synth_lab(1000, H, W).to(dev)
ops.conv_record = []
cc.frame(fr, torch.zeros_like(fr))
rec, ops.conv_record = ops.conv_record, None
uniq = {}
for r in rec:
    k = json.dumps(r, sort_keys=True)
    uniq.setdefault(k, [r, 0])[1] += 1
results = []
tot_auto = tot_best = 0.0
for k, (r, count) in uniq.items():
    x = torch.randn(r["N"], r["Cin"], r["H"], r["W"], device=dev)
    w = torch.randn(r["Cin"], r["ksize"] ** 2, r["Cout"], device=dev) * 0.05
    b = torch.randn(r["Cout"], device=dev)
    OH, OW = ops.conv_out_hw(r["H"], r["W"], r["ksize"], r["stride"], r["dil"], r["pad"], r["in_up"], r["in_sub"])
    sc = sh = sl = res = None
    if r["affine"]:
        sc, sh = torch.rand(r["N"] * r["Cin"], device=dev) + 0.5, torch.randn(r["N"] * r["Cin"], device=dev)
    if r["in_prelu"]:
        sl = torch.tensor([0.25], device=dev)
    if r["residual"]:
        res = torch.randn(r["N"], r["Cout"], OH, OW, device=dev)
    out = torch.empty(r["N"], r["Cout"], OH, OW, device=dev)
    times = {}
    plain = not r["affine"] and not r["in_prelu"]
    for cfg in (-1, 0, 1, 2, 3, 4) + ((16, 17, 18, 19, 20) if plain else ()):   # 16+k: register staging forced
        def run():
            ops.conv2d(x, w, b, ksize=r["ksize"], stride=r["stride"], dil=r["dil"], pad=r["pad"],
                       pad_mode=r["pad_mode"], in_up=r["in_up"], in_sub=r["in_sub"], act=r["act"], act_slope=0.2,
                       in_scale=sc, in_shift=sh, in_slope_t=sl, residual=res, out=out, cfg=cfg)
        try:
            for _ in range(2):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                run()
            e1.record()
            torch.cuda.synchronize()
            times[cfg] = e0.elapsed_time(e1) / 8 * 1e3
        except RuntimeError as e:
            times[cfg] = None
    for sk in (1, 2, 3, 4):
        def run():
            ops.conv2d(x, w, b, ksize=r["ksize"], stride=r["stride"], dil=r["dil"], pad=r["pad"],
                       pad_mode=r["pad_mode"], in_up=r["in_up"], in_sub=r["in_sub"], act=r["act"], act_slope=0.2,
                       in_scale=sc, in_shift=sh, in_slope_t=sl, residual=res, out=out, cfg=-1, split_k=sk)
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            run()
        e1.record()
        torch.cuda.synchronize()
        times[f"sk{sk}"] = e0.elapsed_time(e1) / 8 * 1e3
    # stream-K decomposition: cfg 32 + tile configuration, split_k = workgroups per CU (plain stride-1 layers only)
    if plain and r["stride"] == 1:
        for cfg in (34, 35, 36):
            for per_cu in (1, 2):
                def run():
                    ops.conv2d(x, w, b, ksize=r["ksize"], stride=r["stride"], dil=r["dil"], pad=r["pad"],
                               pad_mode=r["pad_mode"], in_up=r["in_up"], in_sub=r["in_sub"], act=r["act"], act_slope=0.2,
                               residual=res, out=out, cfg=cfg, split_k=per_cu)
                try:
                    for _ in range(2):
                        run()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(8):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    times[f"streamk{cfg - 32}/w{per_cu}"] = e0.elapsed_time(e1) / 8 * 1e3
                except RuntimeError:
                    pass
    flops = 2.0 * r["N"] * r["Cout"] * OH * OW * r["Cin"] * r["ksize"] ** 2
    valid = {c: t for c, t in times.items() if t is not None and c != -1}
    best = min(valid, key=valid.get)
    results.append(dict(shape=r, count=count, OH=OH, OW=OW, gflop=flops / 1e9, us=times, best=best,
                        tflops_auto=flops / times[-1] / 1e6, tflops_best=flops / valid[best] / 1e6))
    tot_auto += count * times[-1]
    tot_best += count * valid[best]
results.sort(key=lambda d: -d["count"] * d["us"][-1])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(H=H, W=W, total_us_auto=tot_auto, total_us_best=tot_best, layers=results),
          open(os.path.join(ROOT, "gpurun_out", "tune_conv.json"), "w"), indent=1)
print(f"conv time per frame: auto {tot_auto / 1e3:.2f} ms, best-per-layer {tot_best / 1e3:.2f} ms")
for d in results[:45]:
    s = d["shape"]
    print(f'{d["count"]}x {"plain" if not s["affine"] and not s["in_prelu"] else "xform"} Cin={s["Cin"]:3d} Cout={s["Cout"]:3d} {s["H"]}x{s["W"]}->{d["OH"]}x{d["OW"]} k{s["ksize"]} s{s["stride"]} d{s["dil"]} '
          f'up{s["in_up"]} sub{s["in_sub"]}: auto {d["us"][-1]:.0f}us ({d["tflops_auto"]:.1f} TF) best {d["best"]} '
          f'{d["us"][d["best"]]:.0f}us ({d["tflops_best"]:.1f} TF) all=' + " ".join(
              f'{c}:{(t if t else 0):.0f}' for c, t in d["us"].items() if c != -1))
