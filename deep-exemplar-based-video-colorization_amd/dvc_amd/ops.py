"""Tensor-level wrappers over the C-ABI (include/dvc_hip.h).

PyTorch is used here only for device memory (torch.empty on the caching allocator) and for the
current HIP stream; every number is produced by the kernels in csrc/.  All functions require
contiguous fp32 ROCm tensors and raise otherwise — there is no CPU fallback.
"""
import ctypes
import sys

import torch

from . import _lib
from ._lib import DvcConvDesc, DvcConvGroupItem, DvcInstNormItem

ACT_NONE, ACT_RELU, ACT_PRELU, ACT_LEAKY, ACT_TANH128 = 0, 1, 2, 3, 4
PAD_ZERO, PAD_REFLECT = 0, 1
EPS64 = sys.float_info.epsilon  # the reference adds float64 epsilon to fp32 norms (util.py:156)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _pv(t):
    """Raw address for a ctypes structure field (None -> NULL)."""
    return None if t is None else t.data_ptr()


# The launch thread issues ~150 kernels per frame and is within 10-15 % of being the bottleneck of the clip driver
# (tools/host_issue_probe.py: 2.1 ms of host time per 2.4 ms frame), so the per-call plumbing uses torch's raw accessors:
# torch.cuda.current_stream() builds a Stream object through four Python frames (3 us, called once or twice per launch).
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _current_device():
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


def _stream_handle():
    """The current HIP stream of the current device as an integer handle."""
    if _raw_stream is not None:
        return _raw_stream(_current_device())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    return ctypes.c_void_p(_stream_handle())


def _need(t, name):
    if t is None:
        return
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"dvc_amd: `{name}` must be a ROCm device tensor; the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"dvc_amd: `{name}` must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError(f"dvc_amd: `{name}` must be contiguous")
    if t.device.index != _current_device():
        # libdvc_hip launches on the CURRENT device's current stream and never calls hipSetDevice:
        # one process per GPU (torch.cuda.set_device(LOCAL_RANK)) is the supported mode
        raise RuntimeError(f"dvc_amd: `{name}` lives on {t.device} but the current device is cuda:"
                           f"{_current_device()}; call torch.cuda.set_device / use torch.cuda.device(...)")


conv_record = None   # set to a list to log every conv2d launch (tools/tune_conv.py)
layer_record = None  # set to a list to log every NAMED 3x3 layer that goes through conv3x3 (tools/engine_sensitivity.py)

# ---- autotuning (the analogue of the reference's `cudnn.benchmark = True`, test.py:140): the first
# time a conv geometry is seen, every (tile configuration, split-K) candidate is timed once on the
# real tensors and the fastest is cached for the rest of the process.  Off by default (static cost
# model in the library); `set_autotune(True)` or DVC_AUTOTUNE=1 turns it on.  Tuning synchronises the
# device, so do it during warm-up, never inside a timed or graph-captured region.
import os as _os
_autotune = _os.environ.get("DVC_AUTOTUNE", "0") == "1"
_tuned = {}


def autotune_enabled():
    return _autotune


def set_autotune(flag=True):
    global _autotune
    _autotune = bool(flag)
    if _autotune:
        _load_tuned()


# optional persistence (DVC_AUTOTUNE_CACHE=<file.json>): tune once, reuse in later processes — e.g. so
# that a profiled run contains production launches only
_cache_path = _os.environ.get("DVC_AUTOTUNE_CACHE")
_cache_loaded = False


def _load_tuned():
    global _cache_loaded
    if _cache_loaded or not _cache_path:
        return
    _cache_loaded = True
    try:
        import json
        with open(_cache_path) as f:
            for k, v in json.load(f).items():
                _tuned[tuple(json.loads(k))] = tuple(v)
    except (OSError, ValueError):
        pass


def _save_tuned():
    if not _cache_path:
        return
    import json
    tmp = _cache_path + ".tmp%d" % _os.getpid()
    with open(tmp, "w") as f:
        json.dump({json.dumps(list(k)): list(v) for k, v in _tuned.items()}, f)
    _os.replace(tmp, _cache_path)


def autotune_table():
    return dict(_tuned)

CONV_WORKSPACE_BYTES = 64 << 20   # split-K scratch: up to 8 partial copies of an under-filled layer's output


def pack_conv_weight(w):
    """[Cout][Cin][kh][kw] -> [Cin][kh*kw][Cout] (layout consumed by dvc_conv2d).  Pure data movement."""
    co, ci, kh, kw = w.shape
    return w.detach().permute(1, 2, 3, 0).reshape(ci, kh * kw, co).contiguous()


def conv_out_hw(H, W, ksize=3, stride=1, dil=1, pad=1, in_up=1, in_sub=1):
    vh = H * 2 if in_up == 2 else ((H + 1) // 2 if in_sub == 2 else H)
    vw = W * 2 if in_up == 2 else ((W + 1) // 2 if in_sub == 2 else W)
    ext = dil * (ksize - 1) + 1
    return (vh + 2 * pad - ext) // stride + 1, (vw + 2 * pad - ext) // stride + 1


def conv2d(x, w_packed, bias, *, ksize=3, stride=1, dil=1, pad=1, pad_mode=PAD_ZERO, in_up=1, in_sub=1,
           act=ACT_NONE, act_slope=0.0, act_slope_t=None, in_scale=None, in_shift=None, in_slope_t=None,
           residual=None, out=None, out_batch_stride=0, cfg=-1, split_k=0, tune=True, gray_input=False):
    """dvc_conv2d.  x: [N,Cin,H,W]; w_packed: [Cin, k*k, Cout].  `out` may be a channel slice view's
    base pointer tensor (pass `out_batch_stride` in elements).  `tune=False`: the library's static plan even with the
    autotuner on (the layers of the error-aware engine map: ONE summation order, the one the parity tests see).
    `gray_input=True` (VGG19 conv1_1 only, DVC_CONV_GRAY_INPUT): x is [N,1,H,W], the centred luminance; the three input
    channels all read (L + 50) / 100 — gray2rgb_batch folded into the load, bit-identical to gray2rgb(x) followed by this call.
    w_packed [N, Cin, k*k, Cout]: per-image filters (DvcConvDesc.w_batch_stride) — with ksize 1 a batched GEMM
    out[n] = w_packed[n]^T x[n], one launch for the whole batch (the training-side N x N products)."""
    lib = _lib.load()
    x_bs = 0
    if gray_input:
        # (the luminance plane is usually the channel-0 slice of a contiguous [N,3,H,W] Lab tensor: rows contiguous, images
        # 3*H*W apart — the descriptor's batch stride carries that, nothing is copied)
        assert x.dim() == 4 and x.shape[1] == 1 and w_packed.dim() == 3 and w_packed.shape[0] == 3 and cfg == -1 and split_k == 0, \
            (x.shape, w_packed.shape)
        if x.stride(3) != 1 or x.stride(2) != x.shape[3]:
            x = x.contiguous()
        x_bs = x.stride(0) if x.shape[0] > 1 else 0
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
            raise RuntimeError("dvc_amd: `x` must be a float32 ROCm device tensor; no CPU fallback")
    for t, nm in ((None if gray_input else x, "x"), (w_packed, "w_packed"), (bias, "bias"), (in_scale, "in_scale"),
                  (in_shift, "in_shift"), (in_slope_t, "in_slope"), (act_slope_t, "act_slope"),
                  (residual, "residual")):
        _need(t, nm)
    N, Cin, H, W = x.shape
    if gray_input:
        Cin, tune = 3, False
    w_bs = 0
    if w_packed.dim() == 4:
        assert w_packed.shape[0] == N, (w_packed.shape, x.shape)
        # (one image: the per-image filter set IS the shared one — no batch stride, so no 16-byte alignment requirement on a
        # filter slice; the one-image-per-call fallbacks of corr_autograd / contextual for odd P*C lean on this)
        w_bs = w_packed[0].numel() if N > 1 else 0
    wshape = tuple(w_packed.shape[-3:])
    assert wshape[0] == Cin and wshape[1] == ksize * ksize, (w_packed.shape, Cin, ksize)
    Cout = wshape[2]
    OH, OW = conv_out_hw(H, W, ksize, stride, dil, pad, in_up, in_sub)
    if out is None:
        out = torch.empty((N, Cout, OH, OW), device=x.device, dtype=torch.float32)
    d = DvcConvDesc(N, Cin, H, W, Cout, ksize, stride, dil, pad, pad_mode, in_up, in_sub, act,
                    float(act_slope), 1 if in_slope_t is not None else 0, cfg, split_k, x_bs, out_batch_stride, 0,
                    GRAY_INPUT if gray_input else 0, w_bs)
    if residual is not None:
        assert tuple(residual.shape) == (N, Cout, OH, OW), (residual.shape, (N, Cout, OH, OW))
    if _autotune and tune and cfg == -1 and split_k == 0:
        # (no N in the key and a single-image descriptor for the timing: the library plans per image, so that a batch is
        # bit-identical to single-image calls — the tuned choice must not depend on the batch size either)
        key = (Cin, H, W, Cout, ksize, stride, dil, pad, pad_mode, in_up, in_sub, in_scale is not None,
               in_slope_t is not None, residual is not None, act, x.device.index, w_packed.dim() == 4)
        best = _tuned.get(key)
        if best is None:
            d1 = DvcConvDesc.from_buffer_copy(d)
            d1.N = 1
            best = _tune_conv(lib, d1, (x, w_packed, bias, in_scale, in_shift, in_slope_t, act_slope_t, residual, out))
            _tuned[key] = best
            _save_tuned()
        d.cfg, d.split_k = best
    if conv_record is not None:
        conv_record.append(dict(N=N, Cin=Cin, H=H, W=W, Cout=Cout, ksize=ksize, stride=stride, dil=dil, pad=pad,
                                pad_mode=pad_mode, in_up=in_up, in_sub=in_sub, affine=in_scale is not None,
                                in_prelu=in_slope_t is not None, residual=residual is not None, act=act))
    ws = _workspace(x.device, CONV_WORKSPACE_BYTES, "conv")
    _bump_generation(ws)
    rc = lib.dvc_conv2d(ctypes.byref(d), _p(x), _p(w_packed), _p(bias), _p(in_scale), _p(in_shift),
                        _p(in_slope_t), _p(act_slope_t), _p(residual), _p(out),
                        ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    _lib.check(rc, "dvc_conv2d")
    return out


# ---- weights-in-registers direct engine (r06, csrc/conv_ws.hip): the large-map / few-channel 3x3 layers that stay on the direct
# engine (ColorVidNet conv1_1[2], conv1_2, conv2_1 under the error-aware map).  DVC_WS_CONV=0: the general direct engine (A/B).
_ws_conv = _os.environ.get("DVC_WS_CONV", "1") != "0"


def ws_conv_enabled():
    return _ws_conv


def set_ws_conv(flag=True):
    global _ws_conv
    _ws_conv = bool(flag)


def ws_eligible(Cin, Cout, dil=1, pad_mode=PAD_ZERO, in_up=1, in_sub=1, act=ACT_NONE):
    """Geometry the weights-in-registers kernel takes (dvc_conv2d_ws_eligible): 3x3 stride 1 pad 1, zero padding, plain input,
    32, 64 or 128 input channels, Cout % 64 == 0."""
    return (Cin in (32, 64, 128) and Cout % 64 == 0 and dil == 1 and pad_mode == PAD_ZERO and in_up == 1 and in_sub == 1
            and act in (ACT_NONE, ACT_RELU, ACT_PRELU, ACT_LEAKY))


def pack_ws_weight(w):
    """[Cout][Cin][3][3] -> the MFMA A-fragment order dvc_conv2d_ws loads (dvc_conv2d_ws_pack_weight), Cout*Cin*9 floats."""
    lib = _lib.load()
    w = w.detach().contiguous().float()
    _need(w, "weight")
    Cout, Cin = w.shape[0], w.shape[1]
    u = torch.empty(Cout * Cin * 9, device=w.device, dtype=torch.float32)
    _lib.check(lib.dvc_conv2d_ws_pack_weight(_p(w), Cout, Cin, _p(u), _stream()), "dvc_conv2d_ws_pack_weight")
    return u


def conv2d_ws(x, u_packed, bias, Cout, *, act=ACT_NONE, act_slope=0.0, act_slope_t=None, out=None, out_batch_stride=0):
    """dvc_conv2d_ws: 3x3 / stride 1 / pad 1 (zero) with the filters resident in registers; see ws_eligible."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (u_packed, "u_packed"), (bias, "bias"), (act_slope_t, "act_slope")):
        _need(t, nm)
    N, Cin, H, W = x.shape
    assert u_packed.numel() == Cout * Cin * 9, (u_packed.shape, Cout, Cin)
    if out is None:
        out = torch.empty((N, Cout, H, W), device=x.device, dtype=torch.float32)
    d = DvcConvDesc(N, Cin, H, W, Cout, 3, 1, 1, 1, PAD_ZERO, 1, 1, act, float(act_slope), 0, -1, 0, 0, out_batch_stride, 0, 0)
    if conv_record is not None:
        conv_record.append(dict(N=N, Cin=Cin, H=H, W=W, Cout=Cout, ksize=3, stride=1, dil=1, pad=1, pad_mode=PAD_ZERO, in_up=1,
                                in_sub=1, affine=False, in_prelu=False, residual=False, act=act, algo="direct-ws"))
    _lib.check(lib.dvc_conv2d_ws(ctypes.byref(d), _p(x), _p(u_packed), _p(bias), _p(act_slope_t), _p(out), _stream()), "dvc_conv2d_ws")
    return out


def pack_winograd_weight(w):
    """[Cout][Cin][3][3] -> the Winograd F(2x2,3x3) transform-domain filters U = G g G^T in the layout
    dvc_conv2d_winograd stages, [Cout/32][Cin][4][32][4] (dvc_winograd_pack_weight: evaluated in double, rounded once)."""
    lib = _lib.load()
    w = w.detach()
    _need(w, "weight")
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (3, 3) and co % 32 == 0, w.shape
    u = torch.empty((co // 32, ci, 4, 32, 4), device=w.device, dtype=torch.float32)
    assert u.numel() == lib.dvc_winograd_weight_floats(co, ci)
    _lib.check(lib.dvc_winograd_pack_weight(_p(w), co, ci, _p(u), _stream()), "dvc_winograd_pack_weight")
    return u


def winograd_eligible(Cin, Cout, ksize=3, stride=1, dil=1, pad=1, in_affine=False, in_prelu=False):
    """Layers dvc_conv2d_winograd takes (include/dvc_hip.h)."""
    return (ksize == 3 and stride == 1 and dil in (1, 2) and pad == dil and not in_affine and not in_prelu
            and Cin % 8 == 0 and Cout % 64 == 0)


DEFER_REDUCE = 1     # DVC_CONV_DEFER_REDUCE
BATCH_PLAN = 2       # DVC_CONV_BATCH_PLAN
GRAY_INPUT = 4       # DVC_CONV_GRAY_INPUT
_batch_plan = False


class batch_plan:
    """While active, Winograd launches that carry a batch are planned for the WHOLE batch (DVC_CONV_BATCH_PLAN): the images'
    workgroups fill the chip together, so under-filled layers drop (part of) their split over input channels — no partial
    sums, no reduce.  Results are deterministic per batch size but no longer bit-identical to single-image calls (the fp32
    summation order over input channels follows the split).  Off by default: everywhere else a batch of N equals N calls bit
    for bit.  Used by the multi-reference pass (ClipColorizer.set_exemplars -> clip), whose R recurrences must run together, and by
    ClipColorizer(batch_plan=True) for batches of independent clips."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global _batch_plan
        self.prev, _batch_plan = _batch_plan, self.on
        return self

    def __exit__(self, *exc):
        global _batch_plan
        _batch_plan = self.prev


def batch_plan_enabled():
    return _batch_plan


def _plan_flags(N):
    return BATCH_PLAN if (_batch_plan and N > 1) else 0
_conv_ws_generation = {}     # convolution workspace (one per device and stream) -> number of convolutions that have used it


def _bump_generation(ws):
    g = _conv_ws_generation.get(ws.data_ptr(), 0) + 1
    _conv_ws_generation[ws.data_ptr()] = g
    return g


class ConvPartials:
    """What conv2d_winograd(..., defer_reduce=True) returns for a layer that is split over input channels: the partial sums
    [S][N][C][H*W] sitting in the stream's convolution workspace, with the bias / activation still to be applied.  Valid
    until the next convolution on the same stream reuses the workspace; the one consumer is instnorm_apply."""

    def __init__(self, ws, S, shape, bias, act, act_slope, act_slope_t, generation, device, offset=0):
        self.ws, self.S, self.shape, self.bias = ws, S, tuple(shape), bias
        self.act, self.act_slope, self.act_slope_t, self.generation, self.device = act, act_slope, act_slope_t, generation, device
        self.offset = int(offset)       # bytes into the workspace (conv3x3_group: several layers' partial sums side by side)

    def data_ptr(self):
        return self.ws.data_ptr() + self.offset

    def check_live(self):
        if self.generation != _conv_ws_generation.get(self.ws.data_ptr()):
            raise RuntimeError("dvc_amd: the convolution workspace holding these partial sums has been reused by a later "
                               "convolution; instnorm_apply must directly follow conv3x3(defer_reduce=True)")


def conv2d_winograd(x, u_packed, bias, *, dil=1, pad_mode=PAD_ZERO, in_up=1, in_sub=1, act=ACT_NONE, act_slope=0.0,
                    act_slope_t=None, residual=None, out=None, out_batch_stride=0, cfg=-1, split_k=0, defer_reduce=False,
                    ws_tag="conv"):
    """dvc_conv2d_winograd: 3x3, stride 1, pad == dil.  u_packed from pack_winograd_weight.
    defer_reduce=True: if the library splits this layer over input channels, skip the reduce launch and return the
    ConvPartials for instnorm_apply to sum (otherwise the ordinary output tensor).
    `ws_tag`: which of the stream's convolution workspaces to use (conv3x3_group's per-layer fall-back keeps the deferred
    partial sums of several layers alive at once)."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (u_packed, "u_packed"), (bias, "bias"), (act_slope_t, "act_slope"), (residual, "residual")):
        _need(t, nm)
    N, Cin, H, W = x.shape
    assert u_packed.dim() == 5 and u_packed.shape[1] == Cin and tuple(u_packed.shape[2:]) == (4, 32, 4), u_packed.shape
    Cout = u_packed.shape[0] * 32
    OH, OW = conv_out_hw(H, W, 3, 1, dil, dil, in_up, in_sub)
    if out is not None:
        defer_reduce = False
    if residual is not None:
        assert tuple(residual.shape) == (N, Cout, OH, OW), (residual.shape, (N, Cout, OH, OW))
    d = DvcConvDesc(N, Cin, H, W, Cout, 3, 1, dil, dil, pad_mode, in_up, in_sub, act, float(act_slope), 0, cfg, split_k,
                    0, out_batch_stride, 0, _plan_flags(N))
    if conv_record is not None:
        conv_record.append(dict(N=N, Cin=Cin, H=H, W=W, Cout=Cout, ksize=3, stride=1, dil=dil, pad=dil, pad_mode=pad_mode,
                                in_up=in_up, in_sub=in_sub, affine=False, in_prelu=False, residual=residual is not None,
                                act=act, algo="winograd"))
    ws = _workspace(x.device, CONV_WORKSPACE_BYTES, ws_tag)
    generation = _bump_generation(ws)
    S = 1
    if defer_reduce and residual is None and OH * OW <= 16384 and act in (ACT_NONE, ACT_RELU, ACT_PRELU, ACT_LEAKY):
        sp, ipl = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(lib.dvc_conv2d_winograd_split(ctypes.byref(d), ws.numel(), ctypes.byref(sp), ctypes.byref(ipl)),
                   "dvc_conv2d_winograd_split")
        S = sp.value
        # (a batch the library would cover in several launches - workspace capacity, 65535-workgroup cap - takes the
        # ordinary reduce: the deferred partial sums must be those of the whole batch)
        if S > 1 and ipl.value >= N and S * N * Cout * OH * OW * 4 <= ws.numel():
            d.flags |= DEFER_REDUCE
        else:
            S = 1
    if S == 1 and out is None:
        out = torch.empty((N, Cout, OH, OW), device=x.device, dtype=torch.float32)
    rc = lib.dvc_conv2d_winograd(ctypes.byref(d), _p(x), _p(u_packed), _p(bias), _p(act_slope_t), _p(residual),
                                 _p(out) if out is not None else ctypes.c_void_p(ws.data_ptr()),
                                 ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    _lib.check(rc, "dvc_conv2d_winograd")
    if S > 1:
        return ConvPartials(ws, S, (N, Cout, OH, OW), bias, act, float(act_slope), act_slope_t, generation, x.device)
    return out


_pool_fusion = _os.environ.get("DVC_POOL_FUSION", "1") != "0"


def pool_fusion():
    return _pool_fusion


def set_pool_fusion(flag=True):
    global _pool_fusion
    _pool_fusion = bool(flag)


def conv2d_winograd_pool(x, u_packed, bias, *, act=ACT_NONE, act_slope=0.0, act_slope_t=None, pad_mode=PAD_ZERO, want_full=True):
    """dvc_conv2d_winograd_pool: (act(conv3x3(x)), maxpool2x2 of it) from one convolution launch (+ its reduce when the layer is
    split); the full-resolution tensor is None with want_full=False.  Bit-identical to conv2d_winograd -> maxpool2x2."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (u_packed, "u_packed"), (bias, "bias"), (act_slope_t, "act_slope")):
        _need(t, nm)
    N, Cin, H, W = x.shape
    assert u_packed.dim() == 5 and u_packed.shape[1] == Cin and tuple(u_packed.shape[2:]) == (4, 32, 4), u_packed.shape
    Cout = u_packed.shape[0] * 32
    d = DvcConvDesc(N, Cin, H, W, Cout, 3, 1, 1, 1, pad_mode, 1, 1, act, float(act_slope), 0, -1, 0, 0, 0, 0, _plan_flags(N))
    if conv_record is not None:
        conv_record.append(dict(N=N, Cin=Cin, H=H, W=W, Cout=Cout, ksize=3, stride=1, dil=1, pad=1, pad_mode=pad_mode,
                                in_up=1, in_sub=1, affine=False, in_prelu=False, residual=False, act=act, algo="winograd"))
    ws = _workspace(x.device, CONV_WORKSPACE_BYTES, "conv")
    _bump_generation(ws)
    full = torch.empty((N, Cout, H, W), device=x.device, dtype=torch.float32) if want_full else None
    pooled = torch.empty((N, Cout, H // 2, W // 2), device=x.device, dtype=torch.float32)
    _lib.check(lib.dvc_conv2d_winograd_pool(ctypes.byref(d), _p(x), _p(u_packed), _p(bias), _p(act_slope_t), _p(full), _p(pooled), 0,
                                            ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
               "dvc_conv2d_winograd_pool")
    return full, pooled


def conv2d_winograd_dual(xA, xB, u_cat, bias, *, dil=1, pad_mode=PAD_ZERO, in_upA=1, in_upB=1, act=ACT_NONE, act_slope=0.0,
                         act_slope_t=None):
    """dvc_conv2d_winograd_dual: act(conv3x3(up_A(xA), W_A) + conv3x3(up_B(xB), W_B) + bias) in one launch.  u_cat = the two
    packed filter sets concatenated along Cin (torch.cat((pack(W_A), pack(W_B)), dim=1)), bias = b_A + b_B."""
    lib = _lib.load()
    for t, nm in ((xA, "xA"), (xB, "xB"), (u_cat, "u_cat"), (bias, "bias"), (act_slope_t, "act_slope")):
        _need(t, nm)
    N, CA, HA, WA = xA.shape
    NB, CB, HB, WB = xB.shape
    assert N == NB and u_cat.dim() == 5 and u_cat.shape[1] == CA + CB and tuple(u_cat.shape[2:]) == (4, 32, 4), (xA.shape, xB.shape, u_cat.shape)
    Cout = u_cat.shape[0] * 32
    OH, OW = conv_out_hw(HA, WA, 3, 1, dil, dil, in_upA, 1)
    if (OH, OW) != conv_out_hw(HB, WB, 3, 1, dil, dil, in_upB, 1):
        raise RuntimeError(f"dvc_amd: conv2d_winograd_dual: the two inputs' virtual sizes differ ({HA * in_upA} x {WA * in_upA} vs "
                           f"{HB * in_upB} x {WB * in_upB})")
    dA = DvcConvDesc(N, CA, HA, WA, Cout, 3, 1, dil, dil, pad_mode, in_upA, 1, act, float(act_slope), 0, -1, 0, 0, 0, 0, _plan_flags(N))
    dB = DvcConvDesc(N, CB, HB, WB, Cout, 3, 1, dil, dil, pad_mode, in_upB, 1, act, float(act_slope), 0, -1, 0, 0, 0, 0, 0)
    if conv_record is not None:
        conv_record.append(dict(N=N, Cin=CA + CB, H=OH, W=OW, Cout=Cout, ksize=3, stride=1, dil=dil, pad=dil, pad_mode=pad_mode,
                                in_up=1, in_sub=1, affine=False, in_prelu=False, residual=False, act=act, algo="winograd-dual"))
    ws = _workspace(xA.device, CONV_WORKSPACE_BYTES, "conv")
    _bump_generation(ws)
    out = torch.empty((N, Cout, OH, OW), device=xA.device, dtype=torch.float32)
    _lib.check(lib.dvc_conv2d_winograd_dual(ctypes.byref(dA), ctypes.byref(dB), _p(xA), _p(xB), _p(u_cat), _p(bias), _p(act_slope_t),
                                            _p(out), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
               "dvc_conv2d_winograd_dual")
    return out


# decoder-block pairs `conv(up(a)) + conv_short(b)` as one launch (DVC_DUAL_CONV=0 / set_dual_conv(False): two launches)
_dual_conv = _os.environ.get("DVC_DUAL_CONV", "1") == "1"


def set_dual_conv(flag=True):
    global _dual_conv
    _dual_conv = bool(flag)


def dual_conv_enabled():
    return _dual_conv


# ---- algorithm choice for the 3x3 stride-1 layers.  "direct": the implicit-GEMM engine everywhere; "winograd": the
# F(2x2,3x3) kernel on every layer it takes; "auto" (default): the static rule below, fitted to
# profiles/r02_conv_wino_probe.txt (Winograd where it is measured faster).  The choice is a pure function of the layer
# geometry, so the clip driver's pipelined and sequential orders still run the same kernels (bit-identical outputs).
_conv_algo = _os.environ.get("DVC_CONV_ALGO", "auto")


# conv -> InstanceNorm pairs: let the InstanceNorm launch sum the convolution's split-K partials (DVC_FUSE_REDUCE=0 disables)
_fuse_reduce = _os.environ.get("DVC_FUSE_REDUCE", "1") == "1"


def fuse_reduce():
    return _fuse_reduce


def set_fuse_reduce(flag=True):
    global _fuse_reduce
    _fuse_reduce = bool(flag)


def set_conv_algo(algo):
    global _conv_algo
    if algo not in ("auto", "speed", "direct", "winograd"):
        raise ValueError("conv algo must be 'auto', 'speed', 'direct' or 'winograd'")
    _conv_algo = algo


def conv_algo():
    return _conv_algo


# the exemplar side of WarpNet memoised behind the reference's unmodified call pattern (nets.WarpNet._memo_exemplar_side).
# DVC_EXEMPLAR_MEMO: "1" (default) on, "0" off, "verify" = on, and every hit ALSO recomputes the exemplar side and compares it
# bit for bit with the memo (a mismatch warns, replaces the memo and returns the fresh value) — the debugging mode for callers
# that write to the exemplar tensors or the WarpNet parameters through `.data`, which no version counter sees.
_exemplar_memo = {"0": "off", "verify": "verify"}.get(_os.environ.get("DVC_EXEMPLAR_MEMO", "1"), "on")


def exemplar_memo_enabled():
    return _exemplar_memo != "off"


def exemplar_memo_mode():
    """"on" | "off" | "verify"."""
    return _exemplar_memo


def set_exemplar_memo(flag=True):
    """True / False, or "verify" (see above)."""
    global _exemplar_memo
    _exemplar_memo = "verify" if flag == "verify" else ("on" if flag else "off")


# ---- error-aware engine map (r05).  Winograd F(2x2,3x3) rounds 2-3x coarser than the direct sum per layer, and through the
# 31 layers of a frame the timed engine ended up FURTHER from the fp64 truth than the reference's own CPU fp32 (r04 review:
# 216x384, plain seed-0 weights, q999 1.8x / max 2.8x the CPU run's).  `tools/engine_sensitivity.py` measures, per named layer,
# what switching that ONE layer from the direct engine to Winograd does to the frame's ab output (the perturbation field, no
# truth needed) and ranks the layers by perturbation energy per microsecond saved; the layers below are the ones "auto" keeps
# on the direct engine so that the whole path is at or below the CPU fp32 run's error against fp64
# (profiles/r05_engine_sensitivity.txt; asserted by tests/test_gpu_nets.py::test_e2e_error_vs_fp64_oracle_next_to_cpu_fp32).
# Names: "vgg.<conv>", "warp.<head>.<index>" / "warp.layer.<b>.conv<k>", "cvn.<key>" (the reference's state_dict prefixes).
# "speed" is the geometry-only rule (what "auto" meant up to r04), "winograd" / "direct" force one engine.
DEFAULT_DIRECT_LAYERS = frozenset(_os.environ["DVC_DIRECT_LAYERS"].split(",")) if _os.environ.get("DVC_DIRECT_LAYERS") is not None else None
_direct_layers = None      # None: the built-in map (arch.DIRECT_LAYERS); a frozenset: an explicit one


def direct_layers():
    """Names of the layers `auto` keeps on the direct engine."""
    if _direct_layers is not None:
        return _direct_layers
    if DEFAULT_DIRECT_LAYERS is not None:
        return DEFAULT_DIRECT_LAYERS
    from . import arch
    return arch.DIRECT_LAYERS


def set_direct_layers(names=None):
    """Replace the error-aware map (None: back to the built-in one).  Captured launch sequences notice (graph._epoch)."""
    global _direct_layers
    _direct_layers = None if names is None else frozenset(names)


def winograd_selected(N, Cin, H, W, Cout, *, ksize=3, stride=1, dil=1, pad=1, in_up=1, in_sub=1, in_affine=False,
                      in_prelu=False, layer=None):
    """True if the current algorithm choice sends this layer to dvc_conv2d_winograd."""
    if _conv_algo == "direct" or not winograd_eligible(Cin, Cout, ksize, stride, dil, pad, in_affine, in_prelu):
        return False
    if _conv_algo == "winograd":
        return True
    if _conv_algo == "auto" and layer is not None and layer in direct_layers():
        return False
    OH, OW = conv_out_hw(H, W, ksize, stride, dil, pad, in_up, in_sub)
    return _wino_rule(N, Cin, Cout, OH, OW, dil)


def _wino_rule(N, Cin, Cout, OH, OW, dil):
    # measured on the MI355X (profiles/r02_conv_algo_sweep.txt): with the two-workgroups-per-CU shape Winograd is faster than
    # or level with the direct engine on every eligible layer of the network down to the 13x24 feature maps
    # (per image, never a function of the batch size: a batch must run the kernels its images would run alone)
    # (r03 sweep: WarpNet's 128 -> 64 at 54x96 is the one eligible layer where the direct engine is level, 21.9 vs 22.5 us as a
    # launch of its own; since r06 it runs inside the grouped launch of the heads' second convolutions, conv3x3_group, which
    # takes Winograd layers only)
    return OH * OW >= 13 * 24


def conv3x3(x, weight, packs, bias, *, dil=1, pad_mode=PAD_ZERO, in_up=1, in_sub=1, act=ACT_NONE, act_slope=0.0,
            act_slope_t=None, residual=None, out=None, out_batch_stride=0, defer_reduce=False, layer=None, ws_tag="conv"):
    """A 3x3 stride-1 pad == dil layer through whichever engine the algorithm choice selects.  `packs(kind)` returns
    the packed weight for kind "direct" ([Cin][9][Cout]) or "winograd" (U = G g G^T), normally from a _PackCache.
    `layer`: the layer's name in the error-aware engine map (direct_layers)."""
    N, Cin, H, W = x.shape
    Cout = weight.shape[0]
    if layer_record is not None:
        layer_record.append(dict(layer=layer, Cin=Cin, Cout=Cout, H=H, W=W, dil=dil, in_up=in_up, in_sub=in_sub,
                                 eligible=winograd_eligible(Cin, Cout, 3, 1, dil, dil)
                                 and _wino_rule(N, Cin, Cout, *conv_out_hw(H, W, 3, 1, dil, dil, in_up, in_sub), dil)))
    if winograd_selected(N, Cin, H, W, Cout, dil=dil, pad=dil, in_up=in_up, in_sub=in_sub, layer=layer):
        return conv2d_winograd(x, packs("winograd"), bias, dil=dil, pad_mode=pad_mode, in_up=in_up, in_sub=in_sub,
                               act=act, act_slope=act_slope, act_slope_t=act_slope_t, residual=residual, out=out,
                               out_batch_stride=out_batch_stride, defer_reduce=defer_reduce and _fuse_reduce, ws_tag=ws_tag)
    if _ws_conv and residual is None and ws_eligible(Cin, Cout, dil, pad_mode, in_up, in_sub, act):
        return conv2d_ws(x, packs("ws"), bias, Cout, act=act, act_slope=act_slope, act_slope_t=act_slope_t, out=out,
                         out_batch_stride=out_batch_stride)
    # (a layer the error-aware map names keeps the library's static plan whatever the autotuner would pick: with the chaotic
    # random weights another split over input channels in ColorVidNet's first layers re-draws the tail of the frame's error
    # field — profiles/r06_parity_pool_probe.txt — and the parity the tests assert must be the parity of the timed run)
    return conv2d(x, packs("direct"), bias, dil=dil, pad=dil, pad_mode=pad_mode, in_up=in_up, in_sub=in_sub, act=act,
                  act_slope=act_slope, act_slope_t=act_slope_t, residual=residual, out=out,
                  out_batch_stride=out_batch_stride, tune=not (_conv_algo == "auto" and layer is not None and layer in direct_layers()))


# ---- the training side's plain batched GEMMs (r06).  The three recompute products of the fused correlation's backward and the
# N x N products of the contextual losses are plain fp32 GEMMs with nothing fused into them; the vendor's library
# (rocBLAS / hipBLASLt behind torch.bmm: fp32 MFMA, exact products, fp32 accumulation — torch's float32 matmul precision is
# "highest" on ROCm and is asserted below) runs them at 103-121 TFLOP/s on the MI355X where this library's 1x1-convolution
# engine reaches 77-81 (tools/gemm_lib_probe.py; the engine keeps the better rounding: blocked sums, 2e-7 against 9e-7 of
# the result's scale — both far inside the tolerances of tests/test_gpu_corr_backward.py).  DVC_GEMM_LIB=0 / set_gemm_lib(False)
# keeps every product on the engine.  The INFERENCE path never comes here.
_gemm_lib = _os.environ.get("DVC_GEMM_LIB", "1") != "0"


def gemm_lib():
    return _gemm_lib


def set_gemm_lib(flag=True):
    global _gemm_lib
    _gemm_lib = bool(flag)


def bmm(a, b, out=None, accumulate=False):
    """out = a @ b (or out += a @ b) for batched fp32 matrices [B, M, K] x [B, K, N] through the vendor GEMM; `a` / `b` may be
    transposed or column-sliced VIEWS (the library takes leading dimensions).  Plain fp32: TF32-like modes are refused."""
    if torch.backends.cuda.matmul.allow_tf32 or torch.get_float32_matmul_precision() != "highest":
        raise RuntimeError("dvc_amd: the training-side GEMMs need torch's float32 matmul precision 'highest' (no TF32); "
                           "restore it, or set DVC_GEMM_LIB=0 to keep these products on this library's own fp32 engine")
    for t, name in ((a, "a"), (b, "b")):
        if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 3:
            raise RuntimeError(f"dvc_amd: `{name}` must be a float32 ROCm tensor [B, M, K]")
    if accumulate:
        if out is None:
            raise RuntimeError("dvc_amd: accumulate needs `out`")
        return torch.baddbmm(out, a, b, out=out)
    return torch.bmm(a, b, out=out) if out is not None else torch.bmm(a, b)


# gray2rgb_batch folded into VGG19 conv1_1's load behind warp_color (r06; DVC_GRAY_FUSION=0: the two launches, bit-identical)
_gray_fusion = _os.environ.get("DVC_GRAY_FUSION", "1") != "0"


def gray_fusion():
    return _gray_fusion


def set_gray_fusion(flag=True):
    global _gray_fusion
    _gray_fusion = bool(flag)


# ---- independent layers as one launch (r06: WarpNet's four heads).  DVC_GROUP_HEADS=0: one launch per layer (A/B; results are
# bit-identical either way — every item keeps the plan and the kernel body it has alone)
_group_heads = _os.environ.get("DVC_GROUP_HEADS", "1") != "0"


def group_heads():
    return _group_heads


def set_group_heads(flag=True):
    global _group_heads
    _group_heads = bool(flag)


def conv3x3_group(items):
    """`items`: list of dicts with the arguments of conv3x3 (x, weight, packs, bias + keywords) for INDEPENDENT layers.  Returns
    the list of their results (tensors, or ConvPartials where `defer_reduce` applies), each bit-identical to its own conv3x3
    call.  One launch for all of them when grouping is on and every item goes to the Winograd engine
    (dvc_conv2d_winograd_group); otherwise the per-layer calls, in order."""
    def single(it, i=0):
        # (per-layer launches: the items' deferred partial sums must all be alive when the caller's next stage consumes them,
        # so every item but the first gets a convolution workspace of its own — same size, hence the same plan)
        kw = {k: v for k, v in it.items() if k not in ("x", "weight", "packs", "bias")}
        return conv3x3(it["x"], it["weight"], it["packs"], it["bias"], ws_tag="conv" if i == 0 else f"conv.g{i}", **kw)

    n = len(items)
    ok = _group_heads and 2 <= n <= 4
    if ok:
        for it in items:
            N, Cin, H, W = it["x"].shape
            dil = it.get("dil", 1)
            if (it.get("out") is not None or it.get("residual") is not None or
                    not winograd_selected(N, Cin, H, W, it["weight"].shape[0], dil=dil, pad=dil, in_up=it.get("in_up", 1),
                                          in_sub=it.get("in_sub", 1), layer=it.get("layer"))):
                ok = False
    if not ok:
        return [single(it, i) for i, it in enumerate(items)]
    lib = _lib.load()
    dev = items[0]["x"].device
    ws = _workspace(dev, CONV_WORKSPACE_BYTES, "conv")
    arr = (DvcConvGroupItem * n)()
    meta = []
    off = 0
    for i, it in enumerate(items):
        x, bias = it["x"], it["bias"]
        u = it["packs"]("winograd")
        act_slope_t = it.get("act_slope_t")
        for t, nm in ((x, "x"), (u, "u_packed"), (bias, "bias"), (act_slope_t, "act_slope")):
            _need(t, nm)
        N, Cin, H, W = x.shape
        Cout = u.shape[0] * 32
        dil, in_up, in_sub = it.get("dil", 1), it.get("in_up", 1), it.get("in_sub", 1)
        act, act_slope = it.get("act", ACT_NONE), float(it.get("act_slope", 0.0))
        OH, OW = conv_out_hw(H, W, 3, 1, dil, dil, in_up, in_sub)
        d = DvcConvDesc(N, Cin, H, W, Cout, 3, 1, dil, dil, it.get("pad_mode", PAD_ZERO), in_up, in_sub, act, act_slope, 0, -1, 0,
                        0, 0, 0, _plan_flags(N))
        if layer_record is not None:
            layer_record.append(dict(layer=it.get("layer"), Cin=Cin, Cout=Cout, H=H, W=W, dil=dil, in_up=in_up, in_sub=in_sub,
                                     eligible=True))
        if conv_record is not None:
            conv_record.append(dict(N=N, Cin=Cin, H=H, W=W, Cout=Cout, ksize=3, stride=1, dil=dil, pad=dil,
                                    pad_mode=it.get("pad_mode", PAD_ZERO), in_up=in_up, in_sub=in_sub, affine=False, in_prelu=False,
                                    residual=False, act=act, algo="winograd"))
        # the split this layer gets ALONE (whole workspace): the grouped launch must reproduce it from this item's share
        sp, ipl = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(lib.dvc_conv2d_winograd_split(ctypes.byref(d), ws.numel(), ctypes.byref(sp), ctypes.byref(ipl)),
                   "dvc_conv2d_winograd_split")
        S = sp.value
        need = (S * N * Cout * OH * OW * 4 + 255) // 256 * 256 if S > 1 else 0
        room = ws.numel() - off
        sp2, ipl2 = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(lib.dvc_conv2d_winograd_split(ctypes.byref(d), max(room, 0), ctypes.byref(sp2), ctypes.byref(ipl2)),
                   "dvc_conv2d_winograd_split")
        if need > room or sp2.value != S or ipl.value < N or ipl2.value < N:
            return [single(it_, j) for j, it_ in enumerate(items)]   # (the workspace cannot hold the items' partial sums side by side)
        defer = (it.get("defer_reduce", False) and _fuse_reduce and S > 1 and OH * OW <= 16384
                 and act in (ACT_NONE, ACT_RELU, ACT_PRELU, ACT_LEAKY))
        if defer:
            d.flags |= DEFER_REDUCE
        out = None if defer else torch.empty((N, Cout, OH, OW), device=dev, dtype=torch.float32)
        a = arr[i]
        a.d = d
        a.x, a.u_packed, a.bias, a.act_slope_ptr, a.residual = x.data_ptr(), u.data_ptr(), _pv(bias), _pv(act_slope_t), None
        a.y = ws.data_ptr() + off if out is None else out.data_ptr()
        a.workspace, a.workspace_bytes = ws.data_ptr() + off, room
        meta.append((out, S, (N, Cout, OH, OW), bias, act, act_slope, act_slope_t, off, u))
        off += need
    generation = _bump_generation(ws)
    _lib.check(lib.dvc_conv2d_winograd_group(arr, n, _stream()), "dvc_conv2d_winograd_group")
    return [out if out is not None else ConvPartials(ws, S, shape, bias, act, act_slope, act_slope_t, generation, dev, offset=o)
            for (out, S, shape, bias, act, act_slope, act_slope_t, o, _u) in meta]


def instnorm_apply_group(items):
    """`items`: list of dicts {x: tensor | ConvPartials, + the keywords of instnorm_apply except `second`} for INDEPENDENT
    norms.  Returns the list of outputs, bit-identical to the per-item instnorm_apply calls; one launch when grouping is on."""
    def single(it):
        kw = {k: v for k, v in it.items() if k != "x"}
        return instnorm_apply(it["x"], **kw)

    n = len(items)
    if not (_group_heads and 2 <= n <= 4):
        return [single(it) for it in items]
    lib = _lib.load()
    arr = (DvcInstNormItem * n)()
    outs = []
    for i, it in enumerate(items):
        x = it["x"]
        part = x if isinstance(x, ConvPartials) else None
        residual, slope_t, chan_scale = it.get("residual"), it.get("slope_t"), it.get("chan_scale")
        up, sub, rpad = it.get("up", 1), it.get("sub", 1), it.get("rpad", 0)
        if part is not None:
            part.check_live()
        for t, nm in ((None if part is not None else x, "x"), (chan_scale, "chan_scale"), (residual, "residual"), (slope_t, "slope")):
            _need(t, nm)
        N, C, H, W = part.shape if part is not None else x.shape
        dev = part.device if part is not None else x.device
        VH, VW = ((H + 1) // 2, (W + 1) // 2) if sub == 2 else (H * up, W * up)
        out = it.get("out")
        if out is None:
            out = torch.empty((N, C, VH + 2 * rpad, VW), device=dev, dtype=torch.float32)
        a = arr[i]
        a.x = part.data_ptr() if part is not None else x.data_ptr()
        a.S = part.S if part is not None else 0
        a.bias = _pv(part.bias) if part is not None else None
        a.act = part.act if part is not None else ACT_NONE
        a.act_slope = part.act_slope if part is not None else 0.0
        a.act_slope_ptr = _pv(part.act_slope_t) if part is not None else None
        a.residual, a.slope_ptr, a.chan_scale = _pv(residual), _pv(slope_t), _pv(chan_scale)
        a.eps = float(it.get("eps", 1e-5))
        a.N, a.C, a.H, a.W, a.up, a.sub, a.rpad = N, C, H, W, up, sub, rpad
        a.x_batch_stride, a.res_batch_stride, a.y_batch_stride = 0, 0, it.get("out_batch_stride", 0)
        a.y = out.data_ptr()
        outs.append(out)
    _lib.check(lib.dvc_instnorm_apply_group(arr, n, _stream()), "dvc_instnorm_apply_group")
    return outs


if _autotune:
    _load_tuned()  # env-enabled autotune: pick up a persisted table


# split-K candidates of the autotuner (DVC_TUNE_SPLITS=1 disables split-K: experiments only)
_TUNE_SPLITS = tuple(int(v) for v in _os.environ.get("DVC_TUNE_SPLITS", "1,2,3,4,6,8").split(","))
_TUNE_STREAMK = tuple((int(a), int(b)) for a, b in (v.split(":") for v in _os.environ.get("DVC_TUNE_STREAMK", "36:2,36:1").split(",") if v))


def _tune_conv(lib, d, tensors):
    """Time every (cfg, split_k) candidate for descriptor `d`; returns the fastest pair."""
    x, w_packed, bias, in_scale, in_shift, in_slope_t, act_slope_t, residual, out = tensors
    if residual is not None and residual.data_ptr() == out.data_ptr():
        # accumulate-in-place call (corr_autograd: d_phi += theta_blk dS): every timing launch would add the product
        # once more into the caller's tensor — time into a scratch output, the caller's launch follows the tuning
        out = torch.empty_like(out)
    ws = _workspace(x.device, CONV_WORKSPACE_BYTES, "conv")
    stream = _stream()

    def launch():
        return lib.dvc_conv2d(ctypes.byref(d), _p(x), _p(w_packed), _p(bias), _p(in_scale), _p(in_shift),
                              _p(in_slope_t), _p(act_slope_t), _p(residual), _p(out),
                              ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream)

    best, best_t = (-1, 0), float("inf")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def time_it():
        launch()
        e0.record()
        for _ in range(3):
            launch()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)

    # the library's own choice first: it is the only way to the kernels that are not a (cfg, split_k) of the general engine
    # (the image-input layers, csrc/conv_image.hip)
    d.cfg, d.split_k = -1, 0
    if launch() == 0:
        best, best_t = (-1, 0), time_it()
    # (layers without a fused input transform stage through LDS-DMA; forcing register staging, cfg 16 + k,
    # never won in the per-layer sweep, so it is not a candidate)
    for cfg in (0, 1, 2, 3, 4):
        for sk in _TUNE_SPLITS:
            d.cfg, d.split_k = cfg, sk
            if launch() != 0:          # configuration does not fit this geometry
                break
            t = time_it()
            if t < best_t:
                best, best_t = (cfg, sk), t
    # stream-K decomposition (cfg 32 + tile configuration; plain stride-1 layers only, the call fails cleanly otherwise):
    # 64x64 tiles, 1 or 2 workgroups per CU
    for cfg, per_cu in _TUNE_STREAMK:
        d.cfg, d.split_k = cfg, per_cu
        if launch() != 0:
            continue
        t = time_it()
        if t < best_t:
            best, best_t = (cfg, per_cu), t
    d.cfg, d.split_k = best
    return best


def conv1x1_small(x, w, bias, act=ACT_NONE):
    lib = _lib.load()
    for t, nm in ((x, "x"), (w, "w"), (bias, "bias")):
        _need(t, nm)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    y = torch.empty((N, Cout, H, W), device=x.device, dtype=torch.float32)
    _lib.check(lib.dvc_conv1x1_small(_p(x), _p(w), _p(bias), N, Cin, H * W, Cout, act, _p(y), _stream()),
               "dvc_conv1x1_small")
    return y


def instnorm_stats(x, eps=1e-5, chan_scale=None):
    """Returns (scale, shift), each [N*C], such that InstanceNorm(x) == x*scale + shift per plane."""
    lib = _lib.load()
    _need(x, "x")
    _need(chan_scale, "chan_scale")
    N, C, H, W = x.shape
    scale = torch.empty(N * C, device=x.device, dtype=torch.float32)
    shift = torch.empty(N * C, device=x.device, dtype=torch.float32)
    _lib.check(lib.dvc_instnorm_stats(_p(x), N, C, H * W, 0, float(eps), _p(chan_scale), _p(scale), _p(shift),
                                      _stream()), "dvc_instnorm_stats")
    return scale, shift


def affine_act(x, scale, shift, *, residual=None, slope_t=None, up=1, rpad=0, out=None, out_batch_stride=0):
    lib = _lib.load()
    for t, nm in ((x, "x"), (scale, "scale"), (shift, "shift"), (residual, "residual"), (slope_t, "slope")):
        _need(t, nm)
    N, C, H, W = x.shape
    if out is None:
        out = torch.empty((N, C, H * up + 2 * rpad, W * up), device=x.device, dtype=torch.float32)
    _lib.check(lib.dvc_affine_act(_p(x), _p(scale), _p(shift), _p(residual), _p(slope_t), N, C, H, W, up, rpad,
                                  0, 0, out_batch_stride, _p(out), _stream()), "dvc_affine_act")
    return out


def instnorm_apply(x, *, eps=1e-5, chan_scale=None, residual=None, slope_t=None, up=1, sub=1, rpad=0, out=None,
                   out_batch_stride=0, second=None):
    """InstanceNorm2d (no affine, biased variance) + optional depthwise scale / skip-add / PReLU / nearest
    upsample / stride-2 subsample / replicate row pad, in one launch; `out=x` normalises in place.
    `second=(chan_scale2, sub2)` also returns InstanceNorm(x) * chan_scale2 at stride sub2 (same statistics,
    same launch): the call then returns (y, y2)."""
    lib = _lib.load()
    part = x if isinstance(x, ConvPartials) else None
    if part is not None:
        part.check_live()
        assert out is None or not isinstance(out, ConvPartials)
        x = None
    for t, nm in ((x, "x"), (chan_scale, "chan_scale"), (residual, "residual"), (slope_t, "slope")):
        _need(t, nm)
    N, C, H, W = part.shape if part is not None else x.shape
    dev = part.device if part is not None else x.device
    VH, VW = ((H + 1) // 2, (W + 1) // 2) if sub == 2 else (H * up, W * up)
    if out is None:
        out = torch.empty((N, C, VH + 2 * rpad, VW), device=dev, dtype=torch.float32)
    cs2, sub2, y2 = None, 1, None
    if second is not None:
        cs2, sub2 = second
        _need(cs2, "chan_scale2")
        y2 = torch.empty((N, C, (H + 1) // 2, (W + 1) // 2) if sub2 == 2 else (N, C, H, W), device=dev,
                         dtype=torch.float32)
    if part is not None:
        _lib.check(lib.dvc_instnorm_apply_partials(ctypes.c_void_p(part.data_ptr()), part.S, _p(part.bias), part.act,
                                                   part.act_slope, _p(part.act_slope_t), _p(residual), _p(slope_t),
                                                   _p(chan_scale), float(eps), N, C, H, W, up, sub, rpad, 0, out_batch_stride,
                                                   _p(out), None, None, _p(cs2), sub2, _p(y2), _stream()),
                   "dvc_instnorm_apply_partials")
        return out if second is None else (out, y2)
    _lib.check(lib.dvc_instnorm_apply(_p(x), _p(residual), _p(slope_t), _p(chan_scale), float(eps), N, C, H, W, up,
                                      sub, rpad, 0, 0, out_batch_stride, _p(out), None, None, _p(cs2), sub2,
                                      _p(y2), _stream()),
               "dvc_instnorm_apply")
    return out if second is None else (out, y2)


def _pool(fn_name, x, k):
    lib = _lib.load()
    _need(x, "x")
    N, C, H, W = x.shape
    y = torch.empty((N, C, H // k, W // k), device=x.device, dtype=torch.float32)
    _lib.check(getattr(lib, fn_name)(_p(x), N * C, H, W, _p(y), _stream()), fn_name)
    return y


def maxpool2x2(x):
    return _pool("dvc_maxpool2x2", x, 2)


def avgpool2x2(x):
    return _pool("dvc_avgpool2x2", x, 2)


def avgpool4x4(x):
    return _pool("dvc_avgpool4x4", x, 4)


def upsample_nearest(x, f):
    lib = _lib.load()
    _need(x, "x")
    N, C, H, W = x.shape
    y = torch.empty((N, C, H * f, W * f), device=x.device, dtype=torch.float32)
    _lib.check(lib.dvc_upsample_nearest(_p(x), N * C, H, W, f, _p(y), _stream()), "dvc_upsample_nearest")
    return y


def channel_l2norm(x, eps=EPS64):
    lib = _lib.load()
    _need(x, "x")
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    _lib.check(lib.dvc_channel_l2norm(_p(x), N, C, H * W, float(eps), _p(y), _stream()), "dvc_channel_l2norm")
    return y


def channel_l2norm_multi(xs, eps=EPS64):
    """feature_normalize of several feature maps of one batch in ONE launch (dvc_channel_l2norm_multi); falls back to one
    launch per map when a map's H*W is not a multiple of 4."""
    lib = _lib.load()
    xs = list(xs)
    for i, x in enumerate(xs):
        _need(x, f"x[{i}]")
    N = xs[0].shape[0]
    # (the one-launch kernel moves float4 pieces: every plane size a multiple of 4 and every base pointer 16-byte aligned —
    # an offset or sliced feature tensor takes the per-map scalar kernel, as it did before the multi entry existed)
    if len(xs) > 8 or any(x.shape[0] != N or (x.shape[2] * x.shape[3]) % 4 or x.data_ptr() % 16 for x in xs):
        return [channel_l2norm(x, eps) for x in xs]
    ys = [torch.empty_like(x) for x in xs]
    n = len(xs)
    px = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
    py = (ctypes.c_void_p * n)(*[y.data_ptr() for y in ys])
    cs = (ctypes.c_int32 * n)(*[x.shape[1] for x in xs])
    hw = (ctypes.c_int32 * n)(*[x.shape[2] * x.shape[3] for x in xs])
    _lib.check(lib.dvc_channel_l2norm_multi(ctypes.cast(px, ctypes.c_void_p), ctypes.cast(py, ctypes.c_void_p),
                                            ctypes.cast(cs, ctypes.c_void_p), ctypes.cast(hw, ctypes.c_void_p), n, N,
                                            float(eps), _stream()), "dvc_channel_l2norm_multi")
    return ys


def gray2rgb(l):
    """l: [N,1,H,W] (may be the channel-0 slice of a contiguous [N,3,H,W] Lab tensor)."""
    lib = _lib.load()
    if not (isinstance(l, torch.Tensor) and l.is_cuda and l.dtype == torch.float32):
        raise RuntimeError("dvc_amd: `l` must be a float32 ROCm device tensor; no CPU fallback")
    N, C, H, W = l.shape
    assert C == 1
    if l.stride(3) != 1 or l.stride(2) != W:
        l = l.contiguous()
    bs = l.stride(0) if N > 1 else H * W
    y = torch.empty((N, 3, H, W), device=l.device, dtype=torch.float32)
    _lib.check(lib.dvc_gray2rgb(_p(l), N, H * W, bs, _p(y), _stream()), "dvc_gray2rgb")
    return y


def lab2rgb(lab, l_offset=0.0):
    lib = _lib.load()
    _need(lab, "lab")
    N, C, H, W = lab.shape
    assert C == 3
    y = torch.empty_like(lab)
    _lib.check(lib.dvc_lab2rgb(_p(lab), N, H * W, float(l_offset), _p(y), _stream()), "dvc_lab2rgb")
    return y


# the merge of the correlation's partial softmax states folded into its consumer (pack_color_input): DVC_FOLD_MERGE=0 / set_fold_merge
_fold_merge = _os.environ.get("DVC_FOLD_MERGE", "1") == "1"


def fold_merge():
    return _fold_merge


def set_fold_merge(flag=True):
    global _fold_merge
    _fold_merge = bool(flag)


class CorrPartials:
    """What corr_fwd(..., defer_merge=True) returns: the per-workgroup partial softmax states of B images (one private
    buffer per image), still to be merged.  The one consumer is pack_color_input, whose launch then merges them and writes the
    warped colours / similarity map straight into ColorVidNet's 7-channel input (dvc_corr_merge_pack) — no warped-Lab /
    similarity tensors, no merge launch, no separate pack launch."""

    def __init__(self, bufs, h, w, temperature):
        self.bufs, self.h, self.w, self.temperature = list(bufs), h, w, float(temperature)
        self.shape = (len(self.bufs), 3, 4 * h, 4 * w)      # of the warped Lab it stands for

    def record_stream(self, stream):
        for b in self.bufs:
            b.record_stream(stream)

    def copy_(self, other):
        for a, b in zip(self.bufs, other.bufs):
            a.copy_(b)
        return self


def _plane(t, ch, name):
    """(pointer, batch stride in elements) of channels ch.. of an [N,C,H,W] fp32 device tensor whose planes are dense
    (a contiguous tensor, or a channel slice of one)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        raise RuntimeError(f"dvc_amd: `{name}` must be a 4-d float32 ROCm device tensor; no CPU fallback")
    if t.device.index != _current_device():
        raise RuntimeError(f"dvc_amd: `{name}` lives on {t.device} but the current device is cuda:{_current_device()}")
    N, C, H, W = t.shape
    if t.stride(3) != 1 or t.stride(2) != W or (C > 1 and t.stride(1) != H * W):
        raise RuntimeError(f"dvc_amd: `{name}` must have dense planes (a contiguous tensor or a channel slice of one)")
    # (an expanded tensor — one frame seen as R images, ClipColorizer._rep — has batch stride 0, which the C-ABI spells -1:
    # 0 is its "densely packed" default)
    return ctypes.c_void_p(t.data_ptr() + 4 * ch * H * W), ((t.stride(0) or -1) if N > 1 else C * H * W)


def pack_color_input(IA_lab, warped_lab, sim, IA_last_lab=None, *, last_l=None, last_ab=None, out=None, want_warped=False):
    """cat((IA_l, warped ab, similarity, IA_last_lab), 1)  (FrameColor.py:63-64).  IA_lab: the current frame (channel 0 is
    read; a Lab tensor or its [:, 0:1] slice).  The previous frame is either `IA_last_lab` [N,3,H,W] or its two parts
    `last_l` (a tensor whose channel 0 is the previous luminance, e.g. the previous Lab frame) and `last_ab` [N,2,H,W] — the
    clip loop passes the parts and never builds test.py:96's cat.  `out`: an existing [N,7,H,W] tensor (graph replay).
    `warped_lab` may be the CorrPartials of a deferred corr_fwd (`sim` is then None): this launch merges them (same arithmetic
    as the merge inside corr_fwd, bit-identical values); want_warped=True additionally materialises the warped Lab
    [N,3,H,W] (what FrameColor.py:41-67 returns) and the call returns (y, warped_lab)."""
    lib = _lib.load()
    if isinstance(warped_lab, CorrPartials):
        return _merge_pack(lib, IA_lab, warped_lab, IA_last_lab, last_l, last_ab, out, want_warped)
    for t, nm in ((warped_lab, "warped_lab"), (sim, "sim")):
        _need(t, nm)
    N, _, H, W = IA_lab.shape
    ia, ia_bs = _plane(IA_lab, 0, "IA_lab")
    if IA_last_lab is not None:
        ll, ll_bs = _plane(IA_last_lab, 0, "IA_last_lab")
        la, la_bs = _plane(IA_last_lab, 1, "IA_last_lab")
    else:
        ll, ll_bs = _plane(last_l, 0, "last_l")
        la, la_bs = _plane(last_ab, 0, "last_ab")
        assert last_ab.shape[1] == 2
    if out is not None:
        _need(out, "out")
        if tuple(out.shape) != (N, 7, H, W) or out.device != IA_lab.device:
            raise RuntimeError(f"dvc_amd: pack_color_input: `out` must be a contiguous float32 [{N}, 7, {H}, {W}] tensor on "
                               f"{IA_lab.device} (got {tuple(out.shape)} on {out.device})")
    y = torch.empty((N, 7, H, W), device=IA_lab.device, dtype=torch.float32) if out is None else out
    _lib.check(lib.dvc_pack_color_input(ia, ia_bs, _p(warped_lab), _p(sim), ll, ll_bs, la, la_bs, N, H * W, _p(y),
                                        _stream()), "dvc_pack_color_input")
    return (y, warped_lab) if want_warped else y


def _merge_pack(lib, IA_lab, part, IA_last_lab, last_l, last_ab, out, want_warped):
    """pack_color_input for the CorrPartials of a deferred corr_fwd: one dvc_corr_merge_pack launch per image."""
    N, _, H, W = IA_lab.shape
    if N != len(part.bufs) or (H, W) != (4 * part.h, 4 * part.w):
        raise RuntimeError(f"dvc_amd: pack_color_input: frame {tuple(IA_lab.shape)} does not fit the correlation's {len(part.bufs)} x "
                           f"{part.h} x {part.w} partial states")
    HW = H * W
    # (the fused launch moves float4 pieces: a view whose planes do not start on 16 bytes — an odd storage offset — is copied
    # once; fresh allocations and whole tensors never are)
    al = lambda t: t if t is None or (t.data_ptr() % 16 == 0 and (t.stride(0) * 4) % 16 == 0) else t.clone(memory_format=torch.contiguous_format)  # noqa: E731
    IA_lab, IA_last_lab, last_l, last_ab = al(IA_lab), al(IA_last_lab), al(last_l), al(last_ab)
    ia, ia_bs = _plane(IA_lab, 0, "IA_lab")
    if IA_last_lab is not None:
        ll, ll_bs = _plane(IA_last_lab, 0, "IA_last_lab")
        la, la_bs = _plane(IA_last_lab, 1, "IA_last_lab")
    else:
        ll, ll_bs = _plane(last_l, 0, "last_l")
        la, la_bs = _plane(last_ab, 0, "last_ab")
        assert last_ab.shape[1] == 2
    if out is not None:
        _need(out, "out")
        if tuple(out.shape) != (N, 7, H, W) or out.device != IA_lab.device:
            raise RuntimeError(f"dvc_amd: pack_color_input: `out` must be a contiguous float32 [{N}, 7, {H}, {W}] tensor on {IA_lab.device}")
    y = torch.empty((N, 7, H, W), device=IA_lab.device, dtype=torch.float32) if out is None else out
    warped = torch.empty((N, 3, H, W), device=IA_lab.device, dtype=torch.float32) if want_warped else None
    step = lambda bs: 0 if bs < 0 else 4 * bs           # noqa: E731  (bytes between images; -1 = the same plane for all)
    st = _stream()
    for n in range(N):
        _lib.check(lib.dvc_corr_merge_pack(ctypes.c_void_p(part.bufs[n].data_ptr()), part.bufs[n].numel(), part.temperature,
                                           part.h, part.w, ctypes.c_void_p(ia.value + n * step(ia_bs)),
                                           ctypes.c_void_p(ll.value + n * step(ll_bs)), ctypes.c_void_p(la.value + n * step(la_bs)),
                                           ctypes.c_void_p(y.data_ptr() + n * 7 * HW * 4),
                                           None if warped is None else ctypes.c_void_p(warped.data_ptr() + n * 3 * HW * 4),
                                           None, st), "dvc_corr_merge_pack")
    return (y, warped) if want_warped else y


def corr_prepare(t_raw, eps=EPS64):
    """t_raw: [B,C,h,w] or [B,C,P] output of the theta/phi 1x1 conv -> centred + normalised [B,C,P]."""
    lib = _lib.load()
    _need(t_raw, "t_raw")
    B, C = t_raw.shape[0], t_raw.shape[1]
    P = t_raw[0, 0].numel()
    out = torch.empty((B, C, P), device=t_raw.device, dtype=torch.float32)
    mean = torch.empty(B * C, device=t_raw.device, dtype=torch.float32)
    _lib.check(lib.dvc_corr_prepare(_p(t_raw), B, C, P, float(eps), _p(mean), _p(out), _stream()),
               "dvc_corr_prepare")
    return out


_ws_cache = {}
_ws_scope = None


class workspace_scope:
    """While active, scratch buffers come from `store` (a dict the caller owns) instead of the per-stream cache: a captured
    launch sequence (dvc_amd/graph.py) bakes its workspace addresses in and may be replayed on any stream, next to eager
    launches or other graphs that use that stream's workspaces — so every capture gets private ones."""

    def __init__(self, store):
        self.store = store

    def __enter__(self):
        global _ws_scope
        self.prev, _ws_scope = _ws_scope, self.store
        return self.store

    def __exit__(self, *exc):
        global _ws_scope
        _ws_scope = self.prev


def _workspace(device, nbytes, tag="corr"):
    if _ws_scope is not None:
        cache, key = _ws_scope, (tag, device.index)
    else:
        cache, key = _ws_cache, (tag, device.index, _stream_handle())
    ws = cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, device=device, dtype=torch.uint8)
        cache[key] = ws
    return ws


def corr_fwd(theta, phi, blab, temperature, h, w, wta_scale=1.0, want_small=False, want_argmax=False,
             want_up=True, defer_merge=False):
    """Fused affinity + softmax + colour gather.  theta/phi: [B,256,P]; blab: [B,3,P] (P = h*w).
    Returns dict with y_up [B,3,4h,4w], sim_up [B,1,4h,4w] and optionally y_small / sim_small / argmax.
    Batch forms: paired (theta, phi, blab all [B]); one exemplar for B frames (phi / blab [1]: the clip driver's batched
    front ends); ONE FRAME AGAINST R EXEMPLARS (theta [1], phi / blab [R]: the references of a clip colourised in one pass,
    test.py:169-181) — outputs then have R images.  The library runs one image per set of launches in every form, so an
    image's result never depends on the form it came in.
    defer_merge=True (wta_scale == 1, no small / arg-max outputs): the merge of the partial softmax states is left to the
    consumer — returns the CorrPartials for pack_color_input instead of the dict."""
    lib = _lib.load()
    for t, nm in ((theta, "theta"), (phi, "phi"), (blab, "blab")):
        _need(t, nm)
    Bt, C, P = theta.shape
    R = phi.shape[0]
    if defer_merge and (wta_scale != 1.0 or want_small or want_argmax):
        defer_merge = False
    shared = Bt > 1 and R == 1 and blab.shape[0] == 1        # one exemplar for a batch of frames (clip driver)
    refs = Bt == 1 and R > 1 and blab.shape[0] == R          # one frame against R exemplars
    B = max(Bt, R)
    assert P == h * w and tuple(phi.shape[1:]) == (C, P) and blab[0].numel() == 3 * P
    assert shared or refs or (R == Bt and blab.shape[0] == Bt), (theta.shape, phi.shape, blab.shape)
    if not (temperature > 0):
        raise ValueError("temperature must be > 0")
    dev = theta.device
    if defer_merge:
        # one private partial-state buffer per image (2 MB at 54x96): the consumer may run on another stream, later
        nb1 = lib.dvc_corr_workspace_bytes(1, P)
        bufs = [torch.empty(nb1, device=dev, dtype=torch.uint8) for _ in range(B)]
        st = _stream()
        for b in range(B):
            th, ph, bl = theta[b if Bt > 1 else 0], phi[b if R > 1 else 0], blab[b if blab.shape[0] > 1 else 0]
            _lib.check(lib.dvc_corr_fwd(_p(th), _p(ph), _p(bl), float(temperature), 1.0, 1, C, h, w, None, None, None, None, None,
                                        ctypes.c_void_p(bufs[b].data_ptr()), bufs[b].numel(), st), "dvc_corr_fwd")
        return CorrPartials(bufs, h, w, temperature)
    out = {}
    if not (want_up or want_small or want_argmax):
        # (at the C-ABI "every output NULL" means a deferred merge: never reach it by accident)
        raise ValueError("dvc_amd: corr_fwd: no output requested (want_up / want_small / want_argmax all False); pass "
                         "defer_merge=True to leave the merge to pack_color_input")
    y_up = sim_up = y_small = sim_small = amax = None
    if want_up:
        y_up = torch.empty((B, 3, 4 * h, 4 * w), device=dev, dtype=torch.float32)
        sim_up = torch.empty((B, 1, 4 * h, 4 * w), device=dev, dtype=torch.float32)
    if want_small:
        y_small = torch.empty((B, 3, h, w), device=dev, dtype=torch.float32)
        sim_small = torch.empty((B, 1, h, w), device=dev, dtype=torch.float32)
    if want_argmax:
        amax = torch.empty((B, P), device=dev, dtype=torch.int32)
    nbytes = lib.dvc_corr_workspace_bytes(1 if (shared or refs) else B, P)
    ws = _workspace(dev, nbytes)

    def call(th, ph, bl, nb, sl):
        o = [None if t is None else t[sl] for t in (y_small, sim_small, y_up, sim_up)]
        rc = lib.dvc_corr_fwd(_p(th), _p(ph), _p(bl), float(temperature), float(wta_scale), nb, C, h, w,
                              _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]),
                              None if amax is None else ctypes.c_void_p(amax[sl].data_ptr()),
                              ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream())
        _lib.check(rc, "dvc_corr_fwd")

    if shared:      # the library runs one image per set of launches anyway (results independent of the batch size)
        for b in range(B):
            call(theta[b:b + 1], phi, blab, 1, slice(b, b + 1))
    elif refs:
        for r in range(R):
            call(theta, phi[r:r + 1], blab[r:r + 1], 1, slice(r, r + 1))
    else:
        call(theta, phi, blab, B, slice(0, B))
    out.update(y_up=y_up, sim_up=sim_up, y_small=y_small, sim_small=sim_small, argmax=amax)
    return out


def corr_prepare_bf16(t_raw, eps=EPS64):
    """t_raw [B,C,h,w] or [B,C,P] -> (fp32 [B,P,C], bf16 [B,P,C] stored as int16) centred + normalised."""
    lib = _lib.load()
    _need(t_raw, "t_raw")
    B, C = t_raw.shape[0], t_raw.shape[1]
    P = t_raw[0, 0].numel()
    f32 = torch.empty((B, P, C), device=t_raw.device, dtype=torch.float32)
    b16 = torch.empty((B, P, C), device=t_raw.device, dtype=torch.int16)
    mean = torch.empty(B * C, device=t_raw.device, dtype=torch.float32)
    _lib.check(lib.dvc_corr_prepare_bf16(_p(t_raw), B, C, P, float(eps), _p(mean), _p(f32),
                                         ctypes.c_void_p(b16.data_ptr()), _stream()), "dvc_corr_prepare_bf16")
    return f32, b16


def corr_fwd_bf16(theta, phi, blab, temperature, h, w, want_small=False, want_argmax=False, want_up=True):
    """bf16 candidate filter + exact fp32 re-scoring.  theta/phi: (fp32 [B,P,C], bf16 [B,P,C]) pairs from
    corr_prepare_bf16; blab [B,3,P].  Same outputs as corr_fwd.  Requires temperature <= 1e-4.
    theta of ONE frame against the R exemplars of phi / blab (multi-reference pass): one set of launches per exemplar."""
    lib = _lib.load()
    (tf, tb), (pf, pb) = theta, phi
    for t, nm in ((tf, "theta"), (pf, "phi"), (blab, "blab")):
        _need(t, nm)
    Bt, P, C = tf.shape
    R = pf.shape[0]
    refs = Bt == 1 and R > 1
    B = R if refs else Bt
    assert P == h * w and (refs or R == Bt) and blab.shape[0] == B
    dev = tf.device
    y_up = sim_up = y_small = sim_small = amax = None
    if want_up:
        y_up = torch.empty((B, 3, 4 * h, 4 * w), device=dev, dtype=torch.float32)
        sim_up = torch.empty((B, 1, 4 * h, 4 * w), device=dev, dtype=torch.float32)
    if want_small:
        y_small = torch.empty((B, 3, h, w), device=dev, dtype=torch.float32)
        sim_small = torch.empty((B, 1, h, w), device=dev, dtype=torch.float32)
    if want_argmax:
        amax = torch.empty((B, P), device=dev, dtype=torch.int32)
    ws = _workspace(dev, lib.dvc_corr_bf16_workspace_bytes(1 if refs else B, P), "corr_bf16")

    def call(tf_, tb_, pf_, pb_, bl, nb, sl):
        o = [None if t is None else t[sl] for t in (y_small, sim_small, y_up, sim_up)]
        rc = lib.dvc_corr_fwd_bf16(ctypes.c_void_p(tb_.data_ptr()), ctypes.c_void_p(pb_.data_ptr()), _p(tf_), _p(pf_),
                                   _p(bl), float(temperature), nb, C, h, w, _p(o[0]), _p(o[1]), _p(o[2]),
                                   _p(o[3]), None if amax is None else ctypes.c_void_p(amax[sl].data_ptr()),
                                   ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream())
        _lib.check(rc, "dvc_corr_fwd_bf16")

    if refs:
        for r in range(R):
            call(tf, tb, pf[r:r + 1], pb[r:r + 1], blab[r:r + 1], 1, slice(r, r + 1))
    else:
        call(tf, tb, pf, pb, blab, B, slice(0, B))
    return dict(y_up=y_up, sim_up=sim_up, y_small=y_small, sim_small=sim_small, argmax=amax)
