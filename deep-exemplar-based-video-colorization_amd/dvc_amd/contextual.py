"""Contextual loss on the HIP kernels — the reference's other N x N cosine-affinity consumer.

`ContextualLoss_forward` / `ContextualLoss` of /root/reference/models/ContextualLoss.py:29-126, as train.py:649-668 calls
them: `contextual_forward_loss(predict_relu5_1, B_relu5_1.detach())` on VGG features of the prediction (gradient wanted)
and of the exemplar (detached).  Same constructor / forward signatures, same per-sample return value [B].

    mu = mean_j Y;  Xn, Yn = (. - mu) / (||.||_C + eps)      dvc_cx_prepare
    S  = Xn^T Yn  in blocks of R rows, the whole batch at once  1x1-convolution engine with per-image filters (ops.conv2d)
    a_i, j*_i, l_i, r_i = max_j A_ij, E_i                     dvc_cx_rows          (A = softmax_j((1 - d / a_i) / h), d = 1 - S)
    column maxima of A over the rows (ContextualLoss only)    dvc_cx_colmax
    loss = -log mean(.)                                       dvc_cx_finish
backward (w.r.t. X; Y is data, as in train.py): S block recomputed, dS by dvc_cx_ds (+ dvc_cx_rows_tq), d Xn = Yn dS^T on
the engine, then dvc_cx_normalize_bwd.  Nothing N x N outlives a row block.  Parity: tests/test_gpu_contextual.py against
float64 autograd through the oracle restatement (oracle/contextual_oracle.py, pinned to the reference module).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib, ops
from .ops import EPS64, _p, _stream

ROW_BLOCK = 2048
BLOCK_BYTES = 256 << 20      # cap of one [B, R, Ny] fp32 buffer (S, and dS^T in the backward pass): R shrinks for big B * Ny


def _row_block(B, Nx, Ny):
    """Rows of S per block: ROW_BLOCK, fewer when the whole batch's block would exceed BLOCK_BYTES (B = 16 on an
    un-downsampled relu3_1, Ny = 5184: 2048 rows would be 0.7 GB per buffer); always a multiple of 64."""
    cap = max(64, (BLOCK_BYTES // (4 * B * Ny)) // 64 * 64)
    return min(ROW_BLOCK, cap, (Nx + 63) // 64 * 64)


def _apply(X, Y, h, centering, mode):
    """The batched launch plan needs 16-byte aligned per-image filters (C * R and Ny * C multiples of 4); odd feature-map
    sizes take one image per call, as up to r03."""
    B, C = X.shape[0], X.shape[1]
    Ny = Y[0, 0].numel()
    if B > 1 and (Ny * C) % 4 != 0:
        return torch.cat([_ContextualCX.apply(X[b:b + 1], Y[b:b + 1], h, centering, mode) for b in range(B)])
    return _ContextualCX.apply(X, Y, h, centering, mode)


def _ip(t):
    return ctypes.c_void_p(t.data_ptr())


class _ContextualCX(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Y, h, centering, mode):
        lib = _lib.load()
        X = X.detach().contiguous().float()
        Y = Y.detach().contiguous().float()
        for t, nm in ((X, "X_features"), (Y, "Y_features")):
            ops._need(t, nm)
        B, C = X.shape[0], X.shape[1]
        Nx, Ny = X[0, 0].numel(), Y[0, 0].numel()
        assert Y.shape[0] == B and Y.shape[1] == C, (X.shape, Y.shape)
        dev = X.device
        f32 = dict(device=dev, dtype=torch.float32)
        Xn, Yn = torch.empty((B, C, Nx), **f32), torch.empty((B, C, Ny), **f32)
        meanY, normX = torch.empty(B * C, **f32), torch.empty((B, Nx), **f32)
        st = _stream()
        _lib.check(lib.dvc_cx_prepare(_p(Y), None, int(centering), B, C, Ny, float(EPS64), _p(meanY), None, _p(Yn), st), "dvc_cx_prepare")
        _lib.check(lib.dvc_cx_prepare(_p(X), _p(meanY), int(centering), B, C, Nx, float(EPS64), None, _p(normX), _p(Xn), st),
                   "dvc_cx_prepare")
        a, l, r, e = (torch.empty((B, Nx), **f32) for _ in range(4))
        jstar = torch.empty((B, Nx), device=dev, dtype=torch.int32)
        cmax = torch.full((B, Ny), -1.0, **f32) if mode == 1 else None
        cargi = torch.zeros((B, Ny), device=dev, dtype=torch.int32) if mode == 1 else None
        loss, gscale = torch.empty(B, **f32), torch.empty(B, **f32)
        R = _row_block(B, Nx, Ny)
        hy, wy = (Y.shape[2], Y.shape[3]) if Y.dim() == 4 else (1, Ny)
        # r04: the whole batch per launch — S[b] = Xn[b]^T Yn[b] is a 1x1 convolution with PER-IMAGE filters (a batched GEMM,
        # DvcConvDesc.w_batch_stride) and the row / column kernels take the batch as a grid dimension
        lib_gemm = ops.gemm_lib()     # r06: S through the vendor's batched GEMM (ops.bmm), no staging copy, no padded rows
        S = None if lib_gemm else torch.empty((B, R, hy, wy), **f32)
        blk = None if lib_gemm else torch.zeros((B, C, 1, R), **f32)           # K-major blocks of Xn columns (one buffer for every block)
        y_img = Yn.view(B, C, hy, wy)
        S_bs = R * Ny
        Sbuf = {}
        for i0 in range(0, Nx, R):
            rows = min(R, Nx - i0)
            if lib_gemm:
                S = Sbuf.setdefault(rows, torch.empty((B, rows, Ny), **f32))
                S_bs = rows * Ny
                ops.bmm(Xn[:, :, i0:i0 + rows].transpose(1, 2), Yn, out=S)
            else:
                _s_block(Xn, y_img, i0, rows, R, S, blk)
            sl = slice(i0, i0 + rows)
            _lib.check(lib.dvc_cx_rows(_p(S), B, S_bs, Nx, rows, Ny, float(h), _p(a[:, sl]), _ip(jstar[:, sl]), _p(l[:, sl]),
                                       _p(r[:, sl]), _p(e[:, sl]), st), "dvc_cx_rows")
            if mode == 1:
                _lib.check(lib.dvc_cx_colmax(_p(S), B, S_bs, Nx, _p(a[:, sl]), _p(l[:, sl]), rows, Ny, i0, float(h), _p(cmax),
                                             _ip(cargi), st), "dvc_cx_colmax")
        v, n = (r, Nx) if mode == 0 else (cmax, Ny)
        _lib.check(lib.dvc_cx_finish(_p(v), B, n, _p(loss), _p(gscale), st), "dvc_cx_finish")
        saved = [Xn, Yn, normX, a, l, r, e, jstar, gscale] + ([cargi] if mode == 1 else [])
        ctx.save_for_backward(*saved)
        ctx.meta = (float(h), int(mode), tuple(X.shape), (hy, wy))
        return loss

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        h, mode, xshape, (hy, wy) = ctx.meta
        saved = ctx.saved_tensors
        Xn, Yn, normX, a, l, r, e, jstar, gscale = saved[:9]
        cargi = saved[9] if mode == 1 else None
        B, C, Nx = Xn.shape
        Ny = Yn.shape[2]
        dev = Xn.device
        f32 = dict(device=dev, dtype=torch.float32)
        st = _stream()
        # the incoming per-sample gradient is folded into the per-sample scale ON THE DEVICE (dvc_cx_ds multiplies the two):
        # no host read-back, the backward pass only enqueues
        gs = (gscale * gout.detach().to(gscale.dtype)).contiguous()
        R = _row_block(B, Nx, Ny)
        lib_gemm = ops.gemm_lib()     # r06: both products through the vendor's batched GEMM (ops.bmm); dS row-major only
        tt, qq = (torch.empty((B, R), **f32), torch.empty((B, R), **f32)) if mode == 1 else (None, None)
        dXn = torch.empty_like(Xn)
        y_img = Yn.view(B, C, hy, wy)
        if lib_gemm:
            bufs = {}
            for i0 in range(0, Nx, R):
                rows = min(R, Nx - i0)
                if rows not in bufs:
                    bufs[rows] = (torch.empty((B, rows, Ny), **f32), torch.empty((B, rows, Ny), **f32), torch.empty((B, C, rows), **f32))
                S, dS, dxb = bufs[rows]
                S_bs = rows * Ny
                ops.bmm(Xn[:, :, i0:i0 + rows].transpose(1, 2), Yn, out=S)
                sl = slice(i0, i0 + rows)
                if mode == 1:
                    _lib.check(lib.dvc_cx_rows_tq(_p(S), B, S_bs, Nx, R, _p(a[:, sl]), _p(l[:, sl]), _ip(cargi), rows, Ny, i0, h, _p(tt),
                                                  _p(qq), st), "dvc_cx_rows_tq")
                _lib.check(lib.dvc_cx_ds(_p(S), B, S_bs, Nx, R, _p(a[:, sl]), _p(l[:, sl]), _p(r[:, sl]), _p(e[:, sl]), _ip(jstar[:, sl]),
                                         None if cargi is None else _ip(cargi), _p(tt), _p(qq), _p(gs), 1.0, mode, rows, Ny, i0, rows, h,
                                         _p(dS), None, st), "dvc_cx_ds")
                # d Xn[b, c, i0 + i] = sum_j Yn[b, c, j] dS[b, i, j]   (one block for all rows: straight into dXn, no copy)
                if rows == Nx:
                    ops.bmm(Yn, dS.transpose(1, 2), out=dXn)
                else:
                    ops.bmm(Yn, dS.transpose(1, 2), out=dxb)
                    dXn[:, :, i0:i0 + rows] = dxb
            dX = torch.empty_like(Xn)
            _lib.check(lib.dvc_cx_normalize_bwd(_p(Xn), _p(normX), _p(dXn), B, C, Nx, float(EPS64), _p(dX), st), "dvc_cx_normalize_bwd")
            return dX.view(xshape), None, None, None, None
        S = torch.empty((B, R, hy, wy), **f32)
        blk = torch.zeros((B, C, 1, R), **f32)
        dST = torch.empty((B, Ny, R // 32, 32), **f32)                       # [Ny][R] per image, as an image of R "pixels"
        y_t = Yn.transpose(1, 2).contiguous().view(B, Ny, 1, C)              # K-major per-image filters of d Xn = Yn dS^T
        S_bs = R * Ny
        for i0 in range(0, Nx, R):
            rows = min(R, Nx - i0)
            _s_block(Xn, y_img, i0, rows, R, S, blk)
            sl = slice(i0, i0 + rows)
            if mode == 1:
                _lib.check(lib.dvc_cx_rows_tq(_p(S), B, S_bs, Nx, R, _p(a[:, sl]), _p(l[:, sl]), _ip(cargi), rows, Ny, i0, h, _p(tt),
                                              _p(qq), st), "dvc_cx_rows_tq")
            _lib.check(lib.dvc_cx_ds(_p(S), B, S_bs, Nx, R, _p(a[:, sl]), _p(l[:, sl]), _p(r[:, sl]), _p(e[:, sl]), _ip(jstar[:, sl]),
                                     None if cargi is None else _ip(cargi), _p(tt), _p(qq), _p(gs), 1.0, mode, rows, Ny, i0, R, h,
                                     None, _p(dST), st), "dvc_cx_ds")
            dxb = ops.conv2d(dST, y_t, None, ksize=1, pad=0)                  # [B, C, R/32, 32]
            dXn[:, :, i0:i0 + rows] = dxb.view(B, C, R)[:, :, :rows]
        dX = torch.empty_like(Xn)
        _lib.check(lib.dvc_cx_normalize_bwd(_p(Xn), _p(normX), _p(dXn), B, C, Nx, float(EPS64), _p(dX), st), "dvc_cx_normalize_bwd")
        return dX.view(xshape), None, None, None, None


def _s_block(Xn, y_img, i0, rows, R, S, blk):
    """S[b, i, :] = sum_c Xn[b, c, i0 + i] Yn[b, c, :] for a block of R rows (zero rows beyond `rows`) of every image of the batch:
    ONE launch of the 1x1-conv engine with per-image filters.  `blk` [B, C, 1, R]: the caller's staging buffer for the blocks'
    columns (stream-ordered reuse, no allocation per block)."""
    if rows < R:
        blk[..., rows:].zero_()
    blk[:, :, 0, :rows].copy_(Xn[:, :, i0:i0 + rows])
    ops.conv2d(y_img, blk, None, ksize=1, pad=0, out=S)


def _check(X_features, Y_features):
    if not (X_features.is_cuda and Y_features.is_cuda):
        raise RuntimeError("ContextualLoss: inputs must be ROCm device tensors; the MI355X HIP path has no CPU fallback")
    if Y_features.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError(
            "ContextualLoss: gradients flow to X_features only (train.py:651-661 passes the exemplar features detached); "
            "detach Y_features")


class ContextualLoss_forward(nn.Module):
    """
    input is Al, Bl, channel = 1, range ~ [0, 255]          (docstring of models/ContextualLoss.py:89-91)
    """

    def __init__(self):
        super().__init__()

    def forward(self, X_features, Y_features, h=0.1, feature_centering=True):
        """X_features & Y_features are feature vectors or feature 2d arrays; h: bandwidth; returns the per-sample loss
        (models/ContextualLoss.py:97-126: CX = mean over X positions of the row maxima of A)."""
        _check(X_features, Y_features)
        return _apply(X_features, Y_features, float(h), bool(feature_centering), 0)


class ContextualLoss(nn.Module):
    """
    input is Al, Bl, channel = 1, range ~ [0, 255]          (docstring of models/ContextualLoss.py:30-32)
    """

    def __init__(self):
        super().__init__()

    def forward(self, X_features, Y_features, h=0.1, feature_centering=True):
        """models/ContextualLoss.py:38-77: CX = mean over Y positions of the column maxima of A."""
        _check(X_features, Y_features)
        return _apply(X_features, Y_features, float(h), bool(feature_centering), 1)
