"""Per-layer rounding error of the two convolution engines on the REAL activations of a frame (GPU box).

tools/engine_sensitivity.py measures how much a change of rounding at layer L moves the frame's ab output; it cannot tell
which engine is the more accurate one at L.  This probe captures every named 3x3 layer's input during one frame and compares,
against an fp64 convolution of the same operands (linear part only: no bias / residual / activation):
    direct     ops.conv2d (library's static choice)
    wino       ops.conv2d_winograd, library's split over input channels
    wino sK    the same with a forced split K (blocked summation: the accumulation chain per transform position is Cin / K)
and times each variant (round-robin, min over rounds).  rel = rms(y - ref) / rms(ref).
Writes gpurun_out/engine_layer_error.{txt,json}."""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from dvc_amd import ops, synth  # noqa: E402
from dvc_amd.frame import ClipColorizer  # noqa: E402
from models.ColorVidNet import ColorVidNet  # noqa: E402
from models.NonlocalNet import VGG19_pytorch, WarpNet  # noqa: E402

H, W = (int(v) for v in os.environ.get("HW", "216x384").split("x"))
SPLITS = [int(v) for v in os.environ.get("SPLITS", "2,4,8").split(",")]
dev = torch.device("cuda")
lines = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    lines.append(s)


with contextlib.redirect_stdout(io.StringIO()):
    nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
for m, s in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
    m.load_state_dict(s)
    m.eval().to(dev)
cc = ClipColorizer(*nets, temperature=1e-10, graph=False)
ops.set_conv_algo("direct")          # capture on one engine; every layer then goes through ops.conv3x3
ops.set_dual_conv(False)
ops.set_pool_fusion(False)
cc.set_exemplar(synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(dev))
fr = synth.synth_lab(synth.FRAME_SEED0, H, W).to(dev)

captured = {}
orig = ops.conv3x3


def capture(x, weight, packs, bias, **kw):
    layer = kw.get("layer")
    if layer is not None and layer not in captured:
        captured[layer] = dict(x=x.detach().clone(), w=weight.detach(), dil=kw.get("dil", 1), pad_mode=kw.get("pad_mode", ops.PAD_ZERO),
                               in_up=kw.get("in_up", 1), in_sub=kw.get("in_sub", 1))
    return orig(x, weight, packs, bias, **kw)


ops.conv3x3 = capture
cc.frame(fr, torch.zeros_like(fr))
ops.conv3x3 = orig
torch.cuda.synchronize()
say(f"{len(captured)} named 3x3 layers captured at {H}x{W}")


def ref64(c):
    x = c["x"].double()
    if c["in_up"] == 2:
        x = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
    if c["in_sub"] == 2:
        x = x[:, :, ::2, ::2]
    d = c["dil"]
    x = F.pad(x, (d, d, d, d), mode="reflect" if c["pad_mode"] == ops.PAD_REFLECT else "constant")
    w = c["w"].double()
    try:
        return F.conv2d(x, w, dilation=d)
    except Exception:      # noqa: BLE001  (no fp64 convolution on this backend: host CPU)
        return F.conv2d(x.cpu(), w.cpu(), dilation=d).to(dev)


def timeit(fn, n=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = []
hdr = f"{'layer':24s} {'geometry':>30s} | {'direct':>9s} {'wino':>9s} " + " ".join(f"{'w s' + str(k):>9s}" for k in SPLITS) + \
      f" | w/d | us: {'direct':>7s} {'wino':>7s} " + " ".join(f"{'w s' + str(k):>7s}" for k in SPLITS)
say(hdr)
for layer, c in captured.items():
    x, w = c["x"], c["w"]
    co, ci = w.shape[0], w.shape[1]
    elig = ops.winograd_eligible(ci, co, 3, 1, c["dil"], c["dil"])
    ref = ref64(c)
    rr = ref.pow(2).mean().sqrt()
    kw = dict(dil=c["dil"], pad_mode=c["pad_mode"], in_up=c["in_up"], in_sub=c["in_sub"])
    wp = ops.pack_conv_weight(w)
    variants = {"direct": lambda: ops.conv2d(x, wp, None, pad=c["dil"], **kw)}
    if elig:
        up = ops.pack_winograd_weight(w)
        variants["wino"] = lambda: ops.conv2d_winograd(x, up, None, **kw)
        for k in SPLITS:
            variants[f"wino_s{k}"] = (lambda k=k: ops.conv2d_winograd(x, up, None, split_k=k, **kw))
    err, us = {}, {}
    for nm, fn in list(variants.items()):
        try:
            y = fn()
            err[nm] = float(((y.double() - ref).pow(2).mean().sqrt() / rr))
        except Exception as e:      # noqa: BLE001  (a forced split that does not fit this geometry)
            err[nm] = None
            variants.pop(nm)
    for _ in range(3):
        for nm, fn in variants.items():
            t = timeit(fn)
            us[nm] = min(us.get(nm, 1e9), t)
    geo = f"{ci}->{co} d{c['dil']} {x.shape[2]}x{x.shape[3]} up{c['in_up']} sub{c['in_sub']}"
    f = lambda v: f"{v:9.2e}" if v is not None else f"{'-':>9s}"      # noqa: E731
    g = lambda v: f"{v:7.1f}" if v is not None else f"{'-':>7s}"      # noqa: E731
    ratio = err["wino"] / err["direct"] if err.get("wino") else float("nan")
    say(f"{layer:24s} {geo:>30s} | {f(err['direct'])} {f(err.get('wino'))} " + " ".join(f(err.get(f'wino_s{k}')) for k in SPLITS) +
        f" | {ratio:4.2f} |     {g(us.get('direct'))} {g(us.get('wino'))} " + " ".join(g(us.get(f'wino_s{k}')) for k in SPLITS))
    rows.append(dict(layer=layer, geometry=geo, err=err, us=us))
    del ref
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "engine_layer_error.json"), "w"), indent=1)
open(os.path.join(ROOT, "gpurun_out", "engine_layer_error.txt"), "w").write("\n".join(lines) + "\n")
