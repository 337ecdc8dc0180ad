"""Differentiable fused correlation — the training-side reuse of the north-star kernel.

The reference trains through `WarpNet.forward` (train.py:402-427 calls frame_colorization under autograd; models/
NonlocalNet.py:477-500): f = theta^T phi, softmax(f / T) with T = 0.01, y = p . B_lab, similarity = max_j f.  Here the
forward is the fused HIP kernel (dvc_corr_fwd) and the backward recomputes the affinities block by block instead of
keeping the P x P matrices autograd would have saved (4 x 107 MB per image at 216x384):

    for each block of R query rows:
        F      = theta_blk^T phi                                   batched GEMM             [R, P]
        dS     = p (g.B_j - g.y_i) / T  (+ d sim at the arg-max)   dvc_corr_softmax_bwd     [R, P] (and [P, R] for the engine)
        d phi += theta_blk dS ;  d theta_blk = phi dS^T            batched GEMMs

The three products are plain fp32 GEMMs: the vendor's batched GEMM by default (r06, ops.bmm), this library's 1x1-convolution
engine with per-image filters under DVC_GEMM_LIB=0 (r04-r05).

Gradients flow to theta and phi (the centred, normalised projections); the pooled exemplar colours are data (no
gradient).  r05: the WTA re-weighting (`WTA_scale_weight != 1`, NonlocalNet.py:288-327; dead in both reference drivers) is
differentiated too, with the reference's own backward rule — factor 1 at the row maximum and the CONSTANT 1e-4 elsewhere,
whatever the scale (NonlocalNet.py:322) — and the similarity map's gradient untouched (it is taken before the re-weighting,
NonlocalNet.py:481-483).  Parity against autograd through the oracle's `correlate`: tests/test_gpu_corr_backward.py.
"""
import ctypes

import torch

from . import _lib, ops
from .ops import _p, _stream

ROW_BLOCK = 2048
BLOCK_BYTES = 768 << 20     # cap of one [images, R, P] fp32 buffer of the backward pass (three of them live at a time)


class _FusedCorrelation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, phi, blab, temperature, h, w, wta_scale=1.0):
        theta, phi, blab = theta.contiguous(), phi.contiguous(), blab.contiguous()
        res = ops.corr_fwd(theta.detach(), phi.detach(), blab.detach(), float(temperature), h, w, wta_scale=float(wta_scale),
                           want_small=True, want_argmax=True, want_up=False)
        y, sim, amax = res["y_small"], res["sim_small"], res["argmax"]
        ctx.save_for_backward(theta.detach(), phi.detach(), blab.detach(), y, sim, amax)
        ctx.temperature, ctx.hw, ctx.wta = float(temperature), (h, w), float(wta_scale)
        ctx.mark_non_differentiable(amax)
        return y, sim, amax

    @staticmethod
    def backward(ctx, gy, gsim, _gamax):
        """r05: the whole batch per launch — the three recompute GEMMs run on the 1x1-convolution engine with PER-IMAGE filters
        (DvcConvDesc.w_batch_stride, as the contextual losses since r04) and dvc_corr_softmax_bwd takes the batch as a grid
        dimension: 5 launches per block of R query rows for B images instead of 5 B (+ the staging copies), and grids B times
        larger.  Images are processed `chunk` at a time so that the three [chunk, R, P] fp32 buffers stay within BLOCK_BYTES."""
        theta, phi, blab, y, sim, amax = ctx.saved_tensors
        T = ctx.temperature
        h, w = ctx.hw
        B, C, P = theta.shape
        lib = _lib.load()
        dev = theta.device
        need_sim = gsim is not None and bool((gsim != 0).any())
        gy = torch.zeros_like(y) if gy is None else gy.contiguous().float()
        gsim_c = gsim.contiguous().float().view(B, P) if need_sim else None
        d_theta = torch.zeros_like(theta)
        d_phi = torch.zeros_like(phi)
        R = min(ROW_BLOCK, (P + 63) // 64 * 64)
        chunk = max(1, min(B, BLOCK_BYTES // (4 * R * P)))
        if (P * C) % 4 or (C * R) % 4:
            chunk = 1               # (per-image filter slices must be 16-byte aligned)
        f32 = dict(device=dev, dtype=torch.float32)
        if ops.gemm_lib():
            return _backward_gemm_lib(ctx, theta, phi, blab, y, sim, amax, gy, gsim_c, need_sim, d_theta, d_phi, R)
        F = torch.empty((chunk, R, h, w), **f32)
        dS = torch.empty((chunk, R, h, w), **f32)
        dST = torch.empty((chunk, P, R // 32, 32), **f32)            # [P][R] per image, as an image of R "pixels"
        lsum = torch.empty(chunk * 3 * R, **f32)                     # row maxima, row sums, raw row maxima of the recomputed blocks
        th_blk = torch.zeros((chunk, C, 1, R), **f32)                # the blocks' theta columns, K-major ...
        th_blk_t = torch.zeros((chunk, R, 1, C), **f32)              # ... and row-major (one pair of buffers for every block)
        yv, gyv, blv, amv = y.view(B, 3, P), gy.view(B, 3, P), blab.view(B, 3, P), amax.view(B, P)
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            sl_b = slice(b0, b0 + nb)
            phi_img = phi[sl_b].view(nb, C, h, w)
            phi_t = phi[sl_b].transpose(1, 2).contiguous().view(nb, P, 1, C)       # K-major per-image filters of d theta = phi dS^T
            dphi_img = d_phi[sl_b].view(nb, C, h, w)
            Fb, dSb, dSTb, tb, tbt = F[:nb], dS[:nb], dST[:nb], th_blk[:nb], th_blk_t[:nb]
            for i0 in range(0, P, R):
                rows = min(R, P - i0)
                if rows < R:
                    tb[..., rows:].zero_()
                    tbt[:, rows:].zero_()
                tb[:, :, 0, :rows].copy_(theta[sl_b, :, i0:i0 + rows])
                tbt[:, :rows, 0, :].copy_(theta[sl_b, :, i0:i0 + rows].transpose(1, 2))
                # F[b, i, :] = sum_c theta[b, c, i0 + i] phi[b, c, :]
                ops.conv2d(phi_img, tb, None, ksize=1, pad=0, out=Fb)
                off = 4 * i0
                rc = lib.dvc_corr_softmax_bwd(
                    _p(Fb), _p(blv[sl_b]), ctypes.c_void_p(gyv[sl_b].data_ptr() + off), ctypes.c_void_p(yv[sl_b].data_ptr() + off),
                    ctypes.c_void_p(sim.view(B, P)[sl_b].data_ptr() + off),
                    ctypes.c_void_p(gsim_c[sl_b].data_ptr() + off) if need_sim else None,
                    ctypes.c_void_p(amv[sl_b].data_ptr() + off) if need_sim else None,
                    T, ctx.wta, nb, rows, P, P, R, _p(lsum), _p(dSb), _p(dSTb), _stream())
                _lib.check(rc, "dvc_corr_softmax_bwd")
                if rows < R:
                    dSb.view(nb, R, P)[:, rows:].zero_()
                # d phi[b, c, j] += sum_i theta[b, c, i0 + i] dS[b, i, j]      (accumulated in place through the skip input)
                ops.conv2d(dSb, tbt, None, ksize=1, pad=0, residual=dphi_img, out=dphi_img)
                # d theta[b, c, i0 + i] = sum_j phi[b, c, j] dS[b, i, j]
                dth = ops.conv2d(dSTb, phi_t, None, ksize=1, pad=0)          # [nb, C, R/32, 32]
                d_theta[sl_b, :, i0:i0 + rows] = dth.view(nb, C, R)[:, :, :rows]
        return d_theta, d_phi, None, None, None, None, None


def _backward_gemm_lib(ctx, theta, phi, blab, y, sim, amax, gy, gsim_c, need_sim, d_theta, d_phi, R):
    """r06: the same block loop with the three products on the vendor's batched GEMM (ops.bmm: 103-121 TFLOP/s against the 1x1
    engine's 77-81 on these shapes).  theta's column block and dS go in as views (no staging copies, no zero padding of the
    last block, no transposed copy of dS: dvc_corr_softmax_bwd(dST = NULL)); d phi accumulates in place (beta = 1)."""
    T = ctx.temperature
    B, C, P = theta.shape
    lib = _lib.load()
    f32 = dict(device=theta.device, dtype=torch.float32)
    chunk = max(1, min(B, BLOCK_BYTES // (4 * R * P)))
    lsum = torch.empty(chunk * 3 * R, **f32)
    yv, gyv, blv, amv = y.view(B, 3, P), gy.view(B, 3, P), blab.view(B, 3, P), amax.view(B, P)
    bufs = {}
    for b0 in range(0, B, chunk):
        nb = min(chunk, B - b0)
        sl_b = slice(b0, b0 + nb)
        for i0 in range(0, P, R):
            rows = min(R, P - i0)
            if (nb, rows) not in bufs:
                bufs[(nb, rows)] = (torch.empty((nb, rows, P), **f32), torch.empty((nb, rows, P), **f32), torch.empty((nb, C, rows), **f32))
            Fb, dSb, dth = bufs[(nb, rows)]
            th_blk = theta[sl_b, :, i0:i0 + rows]                      # [nb, C, rows] view
            ops.bmm(th_blk.transpose(1, 2), phi[sl_b], out=Fb)         # F[b, i, :] = sum_c theta[b, c, i0 + i] phi[b, c, :]
            off = 4 * i0
            rc = lib.dvc_corr_softmax_bwd(
                _p(Fb), _p(blv[sl_b]), ctypes.c_void_p(gyv[sl_b].data_ptr() + off), ctypes.c_void_p(yv[sl_b].data_ptr() + off),
                ctypes.c_void_p(sim.view(B, P)[sl_b].data_ptr() + off),
                ctypes.c_void_p(gsim_c[sl_b].data_ptr() + off) if need_sim else None,
                ctypes.c_void_p(amv[sl_b].data_ptr() + off) if need_sim else None,
                T, ctx.wta, nb, rows, P, P, rows, _p(lsum), _p(dSb), None, _stream())
            _lib.check(rc, "dvc_corr_softmax_bwd")
            ops.bmm(th_blk, dSb, out=d_phi[sl_b], accumulate=True)     # d phi[b, c, j] += sum_i theta[b, c, i0 + i] dS[b, i, j]
            ops.bmm(phi[sl_b], dSb.transpose(1, 2), out=dth)           # d theta[b, c, i0 + i] = sum_j phi[b, c, j] dS[b, i, j]
            d_theta[sl_b, :, i0:i0 + rows] = dth
    return d_theta, d_phi, None, None, None, None, None


def fused_correlation(theta, phi, B_lab_pooled, temperature, h, w, WTA_scale_weight=1):
    """theta, phi: [B, 256, P] centred + L2-normalised projections (what `WarpNet.project` / ops.corr_prepare produce),
    B_lab_pooled: [B, 3, P] = avg_pool2d(B_lab_map, 4) flattened, P = h * w.
    Returns (y [B, 3, h, w], similarity [B, 1, h, w], argmax [B, P]) like the low-resolution stage of WarpNet.forward
    (NonlocalNet.py:477-499 before the x4 nearest upsampling); differentiable w.r.t. theta and phi.
    `WTA_scale_weight` as in WarpNet.forward (NonlocalNet.py:440,486)."""
    return _FusedCorrelation.apply(theta, phi, B_lab_pooled, temperature, h, w, float(WTA_scale_weight))
