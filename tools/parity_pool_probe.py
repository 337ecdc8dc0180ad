"""How far is the timed engine from the fp64 truth, next to the reference's CPU fp32 run — per frame and pooled over several
frames, for several engine configurations (GPU box; the oracle runs on the host CPU: ~10 s per frame for fp32 + fp64).

bench.py's `parity` block and tests/test_gpu_nets.py::test_e2e_error_vs_fp64_oracle_next_to_cpu_fp32 look at ONE frame (seed
1000).  With the plain random weights ColorVidNet amplifies a rounding difference ~70x, so one frame's q99.9 / max of the SAME
arithmetic class moves by tens of percent when only the summation ORDER of a layer changes (another split over input channels,
another engine with the same chain lengths).  This probe separates that lottery from a systematic loss of accuracy.
  python tools/parity_pool_probe.py [--frames 8]      -> gpurun_out/parity_pool_probe.txt"""
import argparse
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dvc_amd import arch, ops, synth  # noqa: E402
from dvc_amd.frame import ClipColorizer  # noqa: E402
from models.ColorVidNet import ColorVidNet  # noqa: E402
from models.NonlocalNet import VGG19_pytorch, WarpNet  # noqa: E402
from oracle import dvc_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
args = ap.parse_args()
H, W, T = 216, 384, 1e-10
dev = torch.device("cuda")
lines = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    lines.append(s)


with contextlib.redirect_stdout(io.StringIO()):
    nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
sd = (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))
for m, s in zip(nets, sd):
    m.load_state_dict(s)
    m.eval().to(dev)
IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W) for i in range(args.frames)]
z = torch.zeros_like(frames[0])
torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
torch.set_flush_denormal(True)
sd64 = tuple(O.to_dtype(s, torch.float64) for s in sd)
t0 = time.time()
with torch.no_grad():
    fB32, fB64 = O.exemplar_features(IB, sd[0]), O.exemplar_features(IB.double(), sd64[0])
    ab32, ab64, w32, w64 = [], [], [], []
    for fr in frames:
        a, w_, _ = O.frame_colorization(fr, IB, z, fB32, *sd, temperature=T)
        ab32.append(a); w32.append(w_)
        a, w_, _ = O.frame_colorization(fr.double(), IB.double(), z.double(), fB64, *sd64, temperature=T)
        ab64.append(a); w64.append(w_)
say(f"oracle fp32 + fp64 on the host CPU, {args.frames} frames: {time.time() - t0:.0f} s")
# frames on which the CPU fp32 arg-max differs from the truth's are a different experiment (a flipped block moves everything)
flip_cpu = [i for i in range(args.frames) if (w32[i].double() - w64[i]).abs().max().item() > 1e-3]
say("CPU fp32 arg-max differs from the fp64 truth on frames", flip_cpu)
st = lambda e: (float(e.max()), float(np.quantile(e.numpy().ravel(), 0.999)), float(e.mean()), float((e ** 2).mean().sqrt()))   # noqa: E731


def evaluate(name, setup):
    setup()
    cc = ClipColorizer(*nets, temperature=T, graph=False)
    cc.set_exemplar(IB.to(dev))
    per, pool_g, pool_c = [], [], []
    for i, fr in enumerate(frames):
        ab, wl = cc.frame(fr.to(dev), z.to(dev))
        if i in flip_cpu or (wl.double().cpu() - w64[i]).abs().max().item() > 1e-3:
            per.append(None)
            continue
        eg, ec = (ab.double().cpu() - ab64[i]).abs(), (ab32[i].double() - ab64[i]).abs()
        g, c = st(eg), st(ec)
        per.append(tuple(g[k] / c[k] for k in range(4)))
        pool_g.append(eg); pool_c.append(ec)
    G, C = st(torch.cat([e.flatten() for e in pool_g])), st(torch.cat([e.flatten() for e in pool_c]))
    say(f"{name:34s} pooled over {len(pool_g)} frames: max {G[0] / C[0]:.2f}  q999 {G[1] / C[1]:.2f}  mean {G[2] / C[2]:.2f}  rms {G[3] / C[3]:.2f}   | per frame "
        "(max/q999/mean): " + "  ".join("flip" if p is None else f"{p[0]:.2f}/{p[1]:.2f}/{p[2]:.2f}" for p in per))


def cfg(ws=True, group=True, tune=False, extra_direct=(), algo="auto", drop=()):
    def f():
        ops.set_conv_algo(algo)
        ops.set_ws_conv(ws)
        ops.set_group_heads(group)
        ops.set_autotune(tune)
        ops.set_direct_layers(None if not (extra_direct or drop) else (frozenset(arch.DIRECT_LAYERS) | frozenset(extra_direct)) - frozenset(drop))
    return f


if os.environ.get("PROBE_MAPS") == "1":     # candidate engine maps (which of the map's layers could go back to Winograd)
    evaluate("default", cfg())
    evaluate("default + autotune", cfg(tune=True))
    evaluate("map - conv3_3", cfg(drop=("cvn.conv3_3",)))
    evaluate("map - conv3_2", cfg(drop=("cvn.conv3_2",)))
    evaluate("map - conv3_2, conv3_3", cfg(drop=("cvn.conv3_2", "cvn.conv3_3")))
    evaluate("map - conv3_2, conv3_3 + autotune", cfg(drop=("cvn.conv3_2", "cvn.conv3_3"), tune=True))
    evaluate("map - conv3_1, conv3_2, conv3_3", cfg(drop=("cvn.conv3_1", "cvn.conv3_2", "cvn.conv3_3")))
    open("gpurun_out/parity_pool_probe_maps.txt", "w").write("\n".join(lines) + "\n")
    sys.exit(0)
evaluate("default (ws, grouped)", cfg())
evaluate("default + autotune (bench)", cfg(tune=True))
evaluate("ws off", cfg(ws=False))
evaluate("ws off + autotune", cfg(ws=False, tune=True))
evaluate("ws off, heads per layer", cfg(ws=False, group=False))
evaluate("ws off, layer3_1.5 direct (r05)", cfg(ws=False, extra_direct=("warp.layer3_1.5",)))
evaluate("ws on, layer3_1.5 direct", cfg(ws=True, extra_direct=("warp.layer3_1.5",)))
evaluate("all direct (ws)", cfg(algo="direct"))
evaluate("all direct (general engine)", cfg(ws=False, algo="direct"))
evaluate("speed (Winograd everywhere)", cfg(algo="speed"))
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/parity_pool_probe.txt", "w").write("\n".join(lines) + "\n")
