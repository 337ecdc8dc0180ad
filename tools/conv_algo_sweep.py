"""Direct engine vs Winograd F(2x2,3x3) on every distinct 3x3 layer of one frame (GPU box).

Records the convolution launches of one frame (ClipColorizer.frame), then times each distinct layer geometry under
  direct   ops.conv2d, library's static tile choice
  wino     ops.conv2d_winograd, library's cost model (position-split kernel, two waves per SIMD)
round-robin after a clock warm-up (min over rounds), and prints per-frame totals for: direct everywhere, Winograd on every
eligible layer, the static rule of ops.winograd_selected ("auto"), and the best of the two per layer.
Writes gpurun_out/conv_algo_sweep.json.   TUNE_H / TUNE_W select the frame size (default 216x384)."""
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
ops.set_conv_algo("direct")
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
ops.set_conv_algo("auto")
uniq = {}
for r in rec:
    r.pop("algo", None)
    uniq.setdefault(json.dumps(r, sort_keys=True), [r, 0])[1] += 1


def timeit(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = []
tot = dict(direct=0.0, wino=0.0, auto=0.0, best=0.0)
gf_tot = 0.0
for k, (r, count) in uniq.items():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(r["N"], r["Cin"], r["H"], r["W"], generator=g).to(dev)
    w = (torch.randn(r["Cout"], r["Cin"], r["ksize"], r["ksize"], generator=g) / (r["Cin"] * r["ksize"] ** 2) ** 0.5).to(dev)
    b = torch.randn(r["Cout"], generator=g).to(dev)
    OH, OW = ops.conv_out_hw(r["H"], r["W"], r["ksize"], r["stride"], r["dil"], r["pad"], r["in_up"], r["in_sub"])
    sc = sh = sl = res = None
    if r["affine"]:
        sc, sh = torch.rand(r["N"] * r["Cin"], device=dev) + 0.5, torch.randn(r["N"] * r["Cin"], device=dev)
    if r["in_prelu"]:
        sl = torch.tensor([0.25], device=dev)
    if r["residual"]:
        res = torch.randn(r["N"], r["Cout"], OH, OW, device=dev)
    wp = ops.pack_conv_weight(w)
    cands = {"direct": lambda: ops.conv2d(x, wp, b, ksize=r["ksize"], stride=r["stride"], dil=r["dil"], pad=r["pad"],
                                          pad_mode=r["pad_mode"], in_up=r["in_up"], in_sub=r["in_sub"], act=r["act"],
                                          act_slope=0.2, in_scale=sc, in_shift=sh, in_slope_t=sl, residual=res)}
    elig = ops.winograd_eligible(r["Cin"], r["Cout"], r["ksize"], r["stride"], r["dil"], r["pad"], r["affine"], r["in_prelu"])
    err = None
    if elig:
        up = ops.pack_winograd_weight(w)
        cands["wino"] = lambda: ops.conv2d_winograd(x, up, b, dil=r["dil"], pad_mode=r["pad_mode"], in_up=r["in_up"],
                                                     in_sub=r["in_sub"], act=r["act"], act_slope=0.2, residual=res)
        yd, yw = cands["direct"](), cands["wino"]()
        err = ((yd - yw).abs().max() / yd.abs().max()).item()
    ws_ok = (r["ksize"] == 3 and r["stride"] == 1 and r["pad"] == 1 and not r["affine"] and not r["in_prelu"] and not r["residual"]
             and ops.ws_eligible(r["Cin"], r["Cout"], r["dil"], r["pad_mode"], r["in_up"], r["in_sub"], r["act"]))
    if ws_ok:       # r06: the weights-in-registers direct engine (csrc/conv_ws.hip)
        uws = ops.pack_ws_weight(w)
        cands["ws"] = lambda: ops.conv2d_ws(x, uws, b, r["Cout"], act=r["act"], act_slope=0.2)
    for _ in range(40):
        cands["direct"]()
    best = {kk: float("inf") for kk in cands}
    for rnd in range(4):
        for kk, f in cands.items():
            best[kk] = min(best[kk], timeit(f))
    sel = elig and ops.winograd_selected(r["N"], r["Cin"], r["H"], r["W"], r["Cout"], ksize=r["ksize"], stride=r["stride"],
                                         dil=r["dil"], pad=r["pad"], in_up=r["in_up"], in_sub=r["in_sub"],
                                         in_affine=r["affine"], in_prelu=r["in_prelu"])
    gf = 2.0 * r["Cin"] * r["Cout"] * r["ksize"] ** 2 * OH * OW * r["N"] / 1e9
    td, tw = best["direct"], best.get("wino", best["direct"])
    tot["direct"] += count * td
    tot["wino"] += count * tw
    tot["auto"] += count * (tw if sel else td)
    tot["best"] += count * min(td, tw)
    gf_tot += count * gf
    rows.append(dict(layer=r, count=count, gflop=gf, us_direct=td, us_direct_ws=best.get("ws"), us_wino=best.get("wino"), auto_is_wino=bool(sel), rel_diff=err))
    print(f"x{count} {r['Cin']:4d}->{r['Cout']:4d} k{r['ksize']} s{r['stride']} d{r['dil']} {r['H']:3d}x{r['W']:3d} up{r['in_up']} sub{r['in_sub']} "
          f"{gf:6.2f} GF: direct {td:6.1f} us" + (f", direct-ws {best['ws']:6.1f} us ({gf / best['ws'] * 1e3:5.1f} TF)" if ws_ok else "") + (f", wino {tw:6.1f} us ({gf / tw * 1e3:5.1f} TF eff), auto={'wino' if sel else 'direct'}, "
                                                   f"|diff|/max = {err:.1e}" if elig else " (not eligible)"), flush=True)
print(f"per frame, {gf_tot:.1f} GFLOP of convolutions: " + ", ".join(f"{k} {v / 1e3:.3f} ms ({gf_tot / v * 1e3:.1f} TF)" for k, v in tot.items()))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(hw=[H, W], totals_us=tot, gflop=gf_tot, layers=rows), open("gpurun_out/conv_algo_sweep.json", "w"), indent=1)
