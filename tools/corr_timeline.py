"""Debug: per-tile phase timeline of corr_fwd_kernel (wave 0 of every workgroup), P=5184."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
os.environ.setdefault("DVC_DEBUG_LIB", "1")   # the dvc_debug_* hooks live in the -DDVC_DEBUG build (make -C csrc DEBUG=1)
import torch  # noqa: E402

from dvc_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.dvc_debug_corr_timeline.restype = None
lib.dvc_debug_corr_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
h, w = 54, 96
P = h * w
th = ops.corr_prepare(torch.randn(1, 256, P, generator=g).to(dev))
ph = ops.corr_prepare(torch.randn(1, 256, P, generator=g).to(dev))
bl = torch.randn(1, 3, P, generator=g).to(dev)
lib.dvc_debug_corr_variant.restype = None
lib.dvc_debug_corr_variant.argtypes = [ctypes.c_int]
for variant in (0, 1):
    lib.dvc_debug_corr_variant(variant)
    for _ in range(3):
        ops.corr_fwd(th, ph, bl, 1e-10, h, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.corr_fwd(th, ph, bl, 1e-10, h, w)
    e1.record()
    torch.cuda.synchronize()
    print(f"variant {variant}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per corr_fwd (+merge)")
lib.dvc_debug_corr_variant(0)
MAXT = 16
nwg = 512
buf = torch.zeros(nwg * MAXT * 4, dtype=torch.int64, device=dev)
lib.dvc_debug_corr_timeline(ctypes.c_void_p(buf.data_ptr()), MAXT)
ops.corr_fwd(th, ph, bl, 1e-10, h, w)
torch.cuda.synchronize()
lib.dvc_debug_corr_timeline(None, 0)
t = buf.view(nwg, MAXT, 4).cpu().double()
valid = t[:, :, 3] > 0
hdr = t[:, MAXT - 1, :]          # entry, loop start, loop end, exit
t = t[:, :MAXT - 1, :]
valid = t[:, :, 3] > 0
base = hdr[:, 0].min()
print("kernel entry  : min 0, median %.0f, max %.0f" % ((hdr[:, 0].median() - base).item(), (hdr[:, 0].max() - base).item()))
print("prologue      : mean %.0f  (min %.0f max %.0f)" % ((hdr[:, 1] - hdr[:, 0]).mean().item(), (hdr[:, 1] - hdr[:, 0]).min().item(), (hdr[:, 1] - hdr[:, 0]).max().item()))
print("tile loop     : mean %.0f  (min %.0f max %.0f)" % ((hdr[:, 2] - hdr[:, 1]).mean().item(), (hdr[:, 2] - hdr[:, 1]).min().item(), (hdr[:, 2] - hdr[:, 1]).max().item()))
print("epilogue      : mean %.0f" % (hdr[:, 3] - hdr[:, 2]).mean().item())
print("exit          : median %.0f, max %.0f  (= kernel span in ticks)" % ((hdr[:, 3].median() - base).item(), (hdr[:, 3].max() - base).item()))
import collections
order = torch.argsort(hdr[:, 0])
print("entry time of wg #0,#128,#255,#256,#300,#400,#491 in dispatch order:", [(hdr[order[i], 0] - base).item() for i in (0, 128, 255, 256, 300, 400, 491)])
t0 = t[:, 0, 0]
print("workgroups:", nwg, "tiles recorded per wg (min/max):", int(valid.sum(1).min()), int(valid.sum(1).max()))
NT = int(valid.sum(1).max()) - 1
full = valid.sum(1) >= NT
tt = t[full][:, :NT]
per_tile = (tt[:, 1:, 0] - tt[:, :-1, 0]).mean().item()
chain = (tt[:, :, 1] - tt[:, :, 0]).mean().item()
fin = (tt[:, :, 2] - tt[:, :, 1]).mean().item()
bar = (tt[:, :, 3] - tt[:, :, 2]).mean().item()
print(f"ticks per tile {per_tile:.1f}: chain(+issue) {chain:.1f}  finish_tile {fin:.1f}  commit+barrier {bar:.1f}")
for k in range(NT):
    print(k, f"chain {(tt[:, k, 1]-tt[:, k, 0]).mean().item():.1f} fin {(tt[:, k, 2]-tt[:, k, 1]).mean().item():.1f} bar {(tt[:, k, 3]-tt[:, k, 2]).mean().item():.1f}")
