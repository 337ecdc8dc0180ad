"""Do graph replays on different HIP streams overlap on the GPU?  Times the pipelined clip driver with every launch issued
from Python, with both per-frame sequences replayed as hipGraphs, and with only one of them replayed; with and without
stream priorities.  GPU box: python tools/graph_overlap_probe.py [trace]   ("trace": one short graph-mode clip only, to be
run under rocprofv3 --kernel-trace; tools/graph_overlap_trace.py reads the CSV)."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")); sys.path.insert(0, ROOT)
import torch
torch.set_num_threads(8)
from dvc_amd import ops, synth
from dvc_amd.frame import ClipColorizer
from models.ColorVidNet import ColorVidNet
from models.NonlocalNet import VGG19_pytorch, WarpNet

H, W = 216, 384
dev = torch.device("cuda")
with contextlib.redirect_stdout(io.StringIO()):
    nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
for m, s in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
    m.load_state_dict(s); m.eval().to(dev)
ops.set_autotune(True)
IB = synth.synth_lab(2, H, W).to(dev)
fr = [synth.synth_lab(1000 + i, H, W).to(dev) for i in range(16)]
mode = sys.argv[1] if len(sys.argv) > 1 else "time"


def make(graph, parts="all", prio=True):
    cc = ClipColorizer(*nets, graph=graph)
    cc.graph_parts = parts
    if not prio:
        cc._main_stream = torch.cuda.Stream()
        cc._side_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    cc.set_exemplar(IB)
    cc.clip(fr[:6], lookahead=2)
    torch.cuda.synchronize()
    return cc


if mode == "trace":
    which = sys.argv[2] if len(sys.argv) > 2 else "graph"
    cc = make(which != "eager", parts="all")
    cc.clip(fr, lookahead=2)
    torch.cuda.synchronize()
    sys.exit(0)

ref = None
for name, kw in (("eager", dict(graph=False)), ("graph: front + color", dict(graph=True, parts="all")),
                 ("graph: front only", dict(graph=True, parts="front")), ("graph: color only", dict(graph=True, parts="color")),
                 ("eager, no stream priorities", dict(graph=False, prio=False)),
                 ("graph: front + color, no stream priorities", dict(graph=True, parts="all", prio=False))):
    cc = make(**kw)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = cc.clip(fr, lookahead=2)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / len(fr))
    if ref is None:
        ref = out
    same = all(torch.equal(a, b) for a, b in zip(out, ref))
    print(f"{name:45s}: {best * 1e3:.3f} ms/frame ({1 / best:.1f} frames/s), bit-identical to eager: {same}", flush=True)
