"""s_memtime stamps of wave 0 of workgroup 0 of the two bf16 correlation passes (-DDVC_DEBUG build): where a pass's time
goes — theta staging, first key tiles, per tile {wait for the tile, barrier, reads + MFMAs + bookkeeping}."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
os.environ.setdefault("DVC_DEBUG_LIB", "1")
import ctypes
import torch
from dvc_amd import _lib, ops
dev = torch.device("cuda")
lib = _lib.load()
for h, w in ((54, 96), (108, 192)):
    P = h * w
    g = torch.Generator().manual_seed(1)
    tr = torch.randn(1, 256, P, generator=g).to(dev); pr = torch.randn(1, 256, P, generator=g).to(dev); bl = torch.randn(1, 3, P, generator=g).to(dev)
    tb, pb = ops.corr_prepare_bf16(tr), ops.corr_prepare_bf16(pr)
    for _ in range(10):
        ops.corr_fwd_bf16(tb, pb, bl, 1e-10, h, w)
    buf = torch.zeros(512, dtype=torch.int64, device=dev)
    lib.dvc_debug_corr_timeline(ctypes.c_void_p(buf.data_ptr()), -1)
    ops.corr_fwd_bf16(tb, pb, bl, 1e-10, h, w)
    torch.cuda.synchronize()
    lib.dvc_debug_corr_timeline(None, -1)
    b = buf.cpu().tolist()
    for ps in (0, 1):
        st = b[ps * 256: ps * 256 + 256]
        n = st[255]
        t = st[:n]
        d = [t[i + 1] - t[i] for i in range(n - 1)]
        print(f"P={P} pass {ps + 1}: {n} stamps, total {t[-1] - t[0]} ticks; entry->theta staged {d[0]}, barrier {d[1]}, ->first tiles/partials landed {d[2]}")
        loop = d[3:-1]
        tiles = [loop[i:i + 3] for i in range(0, len(loop) - 2, 3)]
        print("   per tile [wait vmcnt | barrier | reads+MFMA+sum] (+ bookkeeping to next stamp):", tiles[:6], "...", tiles[-2:])
