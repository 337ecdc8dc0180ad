"""Does the pipelined clip driver get faster when the look-ahead front ends are confined to a subset of the CUs, so that the
ColorVidNet chain (the critical path, high-priority stream) always finds free workgroup slots?  Side streams are created with
hipExtStreamCreateWithCUMask and handed to ClipColorizer as torch ExternalStreams.  GPU box: python tools/cu_mask_probe.py"""
import contextlib, ctypes, io, os, sys, time
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
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")
with contextlib.redirect_stdout(io.StringIO()):
    nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
for m, s in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
    m.load_state_dict(s); m.eval().to(dev)
ops.set_autotune(True)
IB = synth.synth_lab(2, H, W).to(dev)
fr = [synth.synth_lab(1000 + i, H, W).to(dev) for i in range(16)]


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def pattern(kind):
    n = 8           # 256 CUs = 8 words
    if kind == "low half":
        return [0xFFFFFFFF] * 4 + [0] * 4
    if kind == "alternating bits":
        return [0x55555555] * n
    if kind == "alternating words":
        return [0xFFFFFFFF, 0] * 4
    if kind == "three quarters (bits)":
        return [0x77777777] * n
    if kind == "low 16 of every 32":
        return [0x0000FFFF] * n
    raise ValueError(kind)


ref = None
for kind in (None, "low half", "alternating bits", "alternating words", "low 16 of every 32", "three quarters (bits)"):
    for graph in (False, True):
        cc = ClipColorizer(*nets, graph=graph)
        if kind is not None:
            cc._ensure_streams(2)
            cc._side_streams = [masked_stream(pattern(kind)) for _ in range(2)]
        cc.set_exemplar(IB)
        cc.clip(fr[:6], lookahead=2)
        torch.cuda.synchronize()
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
        print(f"side streams: {str(kind):24s} front ends {'replayed' if graph else 'eager   '}: {best * 1e3:.3f} ms/frame ({1 / best:.1f} frames/s), "
              f"bit-identical: {same}", flush=True)
