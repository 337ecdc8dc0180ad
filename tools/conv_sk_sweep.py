"""Per-layer A/B on the GPU box: every distinct convolution of one frame under the tile-per-workgroup engine
(automatic choice and the autotuner's best) and under the stream-K decomposition (cfg 32..36 x 1|2 workgroups per CU).
Writes gpurun_out/conv_sk_sweep.json and prints one line per layer + the per-frame totals."""
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
REPS = int(os.environ.get("SWEEP_REPS", 10))
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


def timeit(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


results = []
tot = {"auto": 0.0, "old_best": 0.0, "sk_best": 0.0, "best": 0.0}
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

    def run(cfg, sk=0):
        ops.conv2d(x, w, b, ksize=r["ksize"], stride=r["stride"], dil=r["dil"], pad=r["pad"], pad_mode=r["pad_mode"],
                   in_up=r["in_up"], in_sub=r["in_sub"], act=r["act"], act_slope=0.2, in_scale=sc, in_shift=sh,
                   in_slope_t=sl, residual=res, out=out, cfg=cfg, split_k=sk)

    times = {"auto": timeit(lambda: run(-1))}
    run(-1)
    ref = out.clone()
    plain = not r["affine"] and not r["in_prelu"] and r["stride"] == 1 and r["Cin"] % (8 if r["ksize"] == 3 else 16) == 0
    for cfg in (4,):
        for skk in (1, 2, 3, 4):
            try:
                times[f"old{cfg}/s{skk}"] = timeit(lambda: run(cfg, skk))
            except RuntimeError:
                pass
    err = {}
    if plain:
        for cfg in (34, 35, 36):
            for per_cu in (1, 2):
                try:
                    times[f"sk{cfg - 32}/w{per_cu}"] = timeit(lambda: run(cfg, per_cu))
                    run(cfg, per_cu)
                    err[f"sk{cfg - 32}/w{per_cu}"] = ((out - ref).abs().max() / ref.abs().max()).item()
                except RuntimeError as e:
                    times[f"sk{cfg - 32}/w{per_cu}"] = None
    flops = 2.0 * r["N"] * r["Cout"] * OH * OW * r["Cin"] * r["ksize"] ** 2
    old = {c: t for c, t in times.items() if c.startswith("old") and t}
    sk = {c: t for c, t in times.items() if c.startswith("sk") and t}
    ob = min(old, key=old.get) if old else None
    sb = min(sk, key=sk.get) if sk else None
    best_t = min([times["auto"]] + ([old[ob]] if ob else []) + ([sk[sb]] if sb else []))
    results.append(dict(shape=r, count=count, OH=OH, OW=OW, gflop=flops / 1e9, us=times, old_best=ob, sk_best=sb,
                        max_rel_diff_vs_auto=err))
    tot["auto"] += count * times["auto"]
    tot["old_best"] += count * (old[ob] if ob else times["auto"])
    tot["sk_best"] += count * (sk[sb] if sb else times["auto"])
    tot["best"] += count * best_t
results.sort(key=lambda d: -d["count"] * d["us"]["auto"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
gf = sum(d["gflop"] * d["count"] for d in results)
json.dump(dict(H=H, W=W, total_us=tot, total_gflop=gf, layers=results),
          open(os.path.join(ROOT, "gpurun_out", "conv_sk_sweep.json"), "w"), indent=1)
print(f"conv GFLOP/frame {gf:.1f}; us/frame: " + ", ".join(f"{k} {v:.0f} ({gf / v * 1e-3:.1f} TF)" for k, v in tot.items()))
for d in results:
    s = d["shape"]
    u = d["us"]
    line = (f'{d["count"]:2d}x Cin={s["Cin"]:3d} Cout={s["Cout"]:3d} {s["H"]}x{s["W"]}->{d["OH"]}x{d["OW"]} k{s["ksize"]} d{s["dil"]} '
            f'up{s["in_up"]} sub{s["in_sub"]}: auto {u["auto"]:.1f}us ({d["gflop"] / u["auto"] * 1e-3:.1f} TF)')
    if d["old_best"]:
        line += f' | old best {d["old_best"]} {u[d["old_best"]]:.1f}'
    if d["sk_best"]:
        line += (f' | SK best {d["sk_best"]} {u[d["sk_best"]]:.1f}us ({d["gflop"] / u[d["sk_best"]] * 1e-3:.1f} TF) all: ' +
                 " ".join(f'{c}:{t:.0f}' for c, t in u.items() if c.startswith("sk") and t) +
                 f' maxdiff {max(d["max_rel_diff_vs_auto"].values()):.1e}')
    print(line)
