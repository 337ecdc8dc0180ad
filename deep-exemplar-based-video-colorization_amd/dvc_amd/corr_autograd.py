"""Differentiable fused correlation — the training-side reuse of the north-star kernel.

The reference trains through `WarpNet.forward` (train.py:402-427 calls frame_colorization under autograd; models/
NonlocalNet.py:477-500): f = theta^T phi, softmax(f / T) with T = 0.01, y = p . B_lab, similarity = max_j f.  Here the
forward is the fused HIP kernel (dvc_corr_fwd) and the backward recomputes the affinities block by block instead of
keeping the P x P matrices autograd would have saved (4 x 107 MB per image at 216x384):

    for each block of R query rows:
        F      = theta_blk^T phi                                   1x1-convolution engine   [R, P]
        dS     = p (g.B_j - g.y_i) / T  (+ d sim at the arg-max)   dvc_corr_softmax_bwd     [R, P] and [P, R]
        d phi += theta_blk dS ;  d theta_blk = phi dS^T            1x1-convolution engine

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
        theta, phi, blab, y, sim, amax = ctx.saved_tensors
        T = ctx.temperature
        h, w = ctx.hw
        B, C, P = theta.shape
        lib = _lib.load()
        dev = theta.device
        need_sim = gsim is not None and bool((gsim != 0).any())
        gy = torch.zeros_like(y) if gy is None else gy.contiguous().float()
        gsim_c = gsim.contiguous().float() if need_sim else None
        d_theta = torch.zeros_like(theta)
        d_phi = torch.zeros_like(phi)
        R = min(ROW_BLOCK, (P + 63) // 64 * 64)
        F = torch.empty((1, R, h, w), device=dev, dtype=torch.float32)
        dS = torch.empty((1, R, h, w), device=dev, dtype=torch.float32)
        dST = torch.empty((1, P, R // 32, 32), device=dev, dtype=torch.float32)     # [P][R] as an image of R "pixels"
        lsum = torch.empty(3 * R, device=dev, dtype=torch.float32)      # row maxima, row sums, raw row maxima of the recomputed block
        th_blk = torch.zeros((C, R), device=dev, dtype=torch.float32)   # the block's theta columns, K-major ...
        th_blk_t = torch.zeros((R, C), device=dev, dtype=torch.float32)  # ... and row-major (one pair of buffers for every block)
        for b in range(B):
            phi_img = phi[b].view(1, C, h, w)
            phi_t = phi[b].t().contiguous().view(P, 1, C)            # K-major weights of d theta = phi dS^T
            dphi_img = d_phi[b].view(1, C, h, w)
            gyb, yb = gy[b].view(3, P), y[b].view(3, P)
            simb = sim[b].view(P)
            for i0 in range(0, P, R):
                rows = min(R, P - i0)
                if rows < R:
                    th_blk[:, rows:].zero_()
                    th_blk_t[rows:].zero_()
                th_blk[:, :rows].copy_(theta[b][:, i0:i0 + rows])
                th_blk_t[:rows].copy_(theta[b][:, i0:i0 + rows].t())
                # F[i, :] = sum_c theta[c, i0 + i] phi[c, :]
                ops.conv2d(phi_img, th_blk.view(C, 1, R), None, ksize=1, pad=0, out=F)
                rc = lib.dvc_corr_softmax_bwd(
                    _p(F), _p(blab[b].view(3, P)), ctypes.c_void_p(gyb.data_ptr() + 4 * i0),
                    ctypes.c_void_p(yb.data_ptr() + 4 * i0), ctypes.c_void_p(simb.data_ptr() + 4 * i0),
                    ctypes.c_void_p(gsim_c[b].view(P).data_ptr() + 4 * i0) if need_sim else None,
                    ctypes.c_void_p(amax[b].data_ptr() + 4 * i0) if need_sim else None,
                    T, ctx.wta, rows, P, P, R, _p(lsum), _p(dS), _p(dST), _stream())
                _lib.check(rc, "dvc_corr_softmax_bwd")
                if rows < R:
                    dS.view(R, P)[rows:].zero_()
                # d phi[c, j] += sum_i theta[c, i0 + i] dS[i, j]      (accumulated in place through the skip input)
                ops.conv2d(dS, th_blk_t.view(R, 1, C), None, ksize=1, pad=0, residual=dphi_img, out=dphi_img)
                # d theta[c, i0 + i] = sum_j phi[c, j] dS[i, j]
                dth = ops.conv2d(dST, phi_t, None, ksize=1, pad=0)          # [1, C, R/32, 32]
                d_theta[b][:, i0:i0 + rows] = dth.view(C, R)[:, :rows]
        return d_theta, d_phi, None, None, None, None, None


def fused_correlation(theta, phi, B_lab_pooled, temperature, h, w, WTA_scale_weight=1):
    """theta, phi: [B, 256, P] centred + L2-normalised projections (what `WarpNet.project` / ops.corr_prepare produce),
    B_lab_pooled: [B, 3, P] = avg_pool2d(B_lab_map, 4) flattened, P = h * w.
    Returns (y [B, 3, h, w], similarity [B, 1, h, w], argmax [B, P]) like the low-resolution stage of WarpNet.forward
    (NonlocalNet.py:477-499 before the x4 nearest upsampling); differentiable w.r.t. theta and phi.
    `WTA_scale_weight` as in WarpNet.forward (NonlocalNet.py:440,486)."""
    return _FusedCorrelation.apply(theta, phi, B_lab_pooled, temperature, h, w, float(WTA_scale_weight))
