"""Clip-driver tail of /root/reference/test.py:98-116 on the GPU: x2 bilinear upsample
of the predicted ab (* 1.25), 8-bit luminance guide, fast global smoother (WLS) and Lab -> 8-bit RGB.

The reference does this on the host per frame (D2H copy, single-core OpenCV-contrib filter, float64 skimage
conversion); here the frame stays in HBM and only the final H x W x 3 uint8 image would leave the device.
"""
import ctypes

import torch

from . import _lib
from .ops import _need, _p, _stream, _workspace, avgpool2x2


def upsample_ab(ab, mul=1.25):
    """torch.nn.functional.interpolate(ab, scale_factor=2, mode="bilinear") * 1.25   (test.py:100-102)."""
    lib = _lib.load()
    _need(ab, "ab")
    N, C, H, W = ab.shape
    y = torch.empty((N, C, 2 * H, 2 * W), device=ab.device, dtype=torch.float32)
    _lib.check(lib.dvc_upsample_bilinear2x(_p(ab), N * C, H, W, float(mul), _p(y), _stream()),
               "dvc_upsample_bilinear2x")
    return y


def downsample_half(x):
    """torch.nn.functional.interpolate(x, scale_factor=0.5, mode="bilinear")   (test.py:58,71) — which ATen
    evaluates exactly as a 2x2 average pool (checked bit for bit in tests/test_tail.py)."""
    return avgpool2x2(x)


def luminance_guide_u8(L_centered):
    """(uncenter_l(L) * 255 / 100).astype(np.uint8)   (test.py:106-109); any shape, returns uint8 of that shape."""
    lib = _lib.load()
    _need(L_centered, "L")
    g = torch.empty(L_centered.shape, device=L_centered.device, dtype=torch.uint8)
    _lib.check(lib.dvc_lum_guide_u8(_p(L_centered), L_centered.numel(), ctypes.c_void_p(g.data_ptr()), _stream()),
               "dvc_lum_guide_u8")
    return g


def fgs_filter(guide_u8, src, lambda_value=500.0, sigma_color=4.0, num_iter=3, lambda_attenuation=0.25, second_coeff_copy=False):
    """cv2.ximgproc.createFastGlobalSmootherFilter(guide, lambda, sigma_color).filter(plane) (test.py:107-111).
    One frame: guide_u8 [H, W] uint8, src [planes, H, W].  Several frames in one call: guide_u8 [G, H, W],
    src [G, planes, H, W] (every frame's planes are filtered with that frame's guide).
    `second_coeff_copy`: hand the library room for a second copy of the elimination coefficients — only a lambda beyond the
    windowed coefficient kernel's reach (> ~4e3) uses it (coefficients by thread-per-line chains, transposed for the scan
    solver); without it that case runs the thread-per-line solver.  The library is always told exactly the size asked for
    here, never what an earlier, larger call left in the workspace cache: the path taken does not depend on history."""
    lib = _lib.load()
    _need(src, "src")
    if guide_u8.dtype != torch.uint8 or not guide_u8.is_cuda or not guide_u8.is_contiguous():
        raise RuntimeError("dvc_amd: `guide` must be a contiguous uint8 ROCm tensor")
    batched = guide_u8.dim() == 3
    G = guide_u8.shape[0] if batched else 1
    ppg = src.shape[1] if batched else src.shape[0]
    H, W = src.shape[-2:]
    if tuple(guide_u8.shape[-2:]) != (H, W) or (batched and (src.dim() != 4 or src.shape[0] != G)):
        raise RuntimeError(f"dvc_amd: guide {tuple(guide_u8.shape)} does not fit src {tuple(src.shape)}")
    dst = torch.empty_like(src)
    need = lib.dvc_fgs_workspace_bytes(H, W, G, ppg, int(num_iter))
    if second_coeff_copy:
        need += 4 * 6 * int(num_iter) * G * H * W
    ws = _workspace(src.device, need, "fgs")
    _lib.check(lib.dvc_fgs_filter(ctypes.c_void_p(guide_u8.data_ptr()), _p(src), G, ppg, H, W, float(lambda_value),
                                  float(sigma_color), int(num_iter), float(lambda_attenuation), _p(dst),
                                  ctypes.c_void_p(ws.data_ptr()), need, _stream()), "dvc_fgs_filter")
    return dst


def lab_to_rgb8(L_centered, ab):
    """batch_lab2rgb_transpose_mc(L[:1], ab[:1]) (utils/util.py:134-151): [H,W] + [2,H,W] -> uint8 [H,W,3]."""
    lib = _lib.load()
    _need(L_centered, "L")
    _need(ab, "ab")
    H, W = L_centered.shape[-2:]
    rgb = torch.empty((H, W, 3), device=ab.device, dtype=torch.uint8)
    _lib.check(lib.dvc_lab2rgb_u8(_p(L_centered), _p(ab), H, W, ctypes.c_void_p(rgb.data_ptr()), _stream()),
               "dvc_lab2rgb_u8")
    return rgb


def rgb8_to_lab(rgb_hwc):
    """Normalize()(ToTensor()(RGB2Lab()(image))) (utils/util_distortion.py:18-23,85-100) for an 8-bit H x W x 3
    device image: returns the centred Lab tensor [1, 3, H, W] that test.py:70 calls `IA_lab_large`."""
    lib = _lib.load()
    if rgb_hwc.dtype != torch.uint8 or not rgb_hwc.is_cuda or not rgb_hwc.is_contiguous() or rgb_hwc.shape[-1] != 3:
        raise RuntimeError("dvc_amd: `rgb` must be a contiguous uint8 ROCm tensor [H, W, 3]")
    H, W = rgb_hwc.shape[:2]
    lab = torch.empty((1, 3, H, W), device=rgb_hwc.device, dtype=torch.float32)
    _lib.check(lib.dvc_rgb8_to_lab(ctypes.c_void_p(rgb_hwc.data_ptr()), H, W, _p(lab), _stream()), "dvc_rgb8_to_lab")
    return lab


def center_pad(rgb_hwc, image_size, three_pass=False):
    """CenterPad(image_size)(image) (utils/util_distortion.py:217-258) for an 8-bit H0 x W0 x 3 device image:
    anti-aliased resize to the target width / height, centre crop -> uint8 [H, W, 3].
    One fused launch where it applies (down-scaling by up to 3.25: r06); `three_pass=True` forces the full-frame passes of
    r01 (the same bytes; what every other factor runs)."""
    lib = _lib.load()
    if rgb_hwc.dtype != torch.uint8 or not rgb_hwc.is_cuda or not rgb_hwc.is_contiguous() or rgb_hwc.shape[-1] != 3:
        raise RuntimeError("dvc_amd: `rgb` must be a contiguous uint8 ROCm tensor [H, W, 3]")
    H0, W0 = rgb_hwc.shape[:2]
    H, W = int(image_size[0]), int(image_size[1])
    out = torch.empty((H, W, 3), device=rgb_hwc.device, dtype=torch.uint8)
    if not three_pass and lib.dvc_center_pad_is_fused(H0, W0, H, W):
        _lib.check(lib.dvc_center_pad(ctypes.c_void_p(rgb_hwc.data_ptr()), H0, W0, H, W, ctypes.c_void_p(out.data_ptr()),
                                      None, 0, _stream()), "dvc_center_pad")
        return out
    ws = _workspace(rgb_hwc.device, lib.dvc_center_pad_workspace_bytes(H0, W0), "ingest")
    _lib.check(lib.dvc_center_pad(ctypes.c_void_p(rgb_hwc.data_ptr()), H0, W0, H, W, ctypes.c_void_p(out.data_ptr()),
                                  ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "dvc_center_pad")
    return out


def frame_ingest(rgb_hwc, image_size):
    """transform(frame) of test.py:44-46 on the device: CenterPad -> (CenterCrop: identity) -> RGB2Lab -> ToTensor
    -> Normalize; uint8 [H0, W0, 3] -> centred Lab [1, 3, H, W] (`IA_lab_large`, test.py:70)."""
    return rgb8_to_lab(center_pad(rgb_hwc, image_size))


def frame_tail(IA_lab_large, I_current_ab_predict, wls_filter_on=True, lambda_value=500, sigma_color=4):
    """test.py:98-116 for one frame (batch 1): returns (IA_predict_rgb uint8 [2H,2W,3] on the device,
    curr_predict[_filter] float32 [1,2,2H,2W]).  Names follow the reference."""
    if IA_lab_large.shape[0] != 1 or I_current_ab_predict.shape[0] != 1:
        raise RuntimeError("frame_tail: batch 1 only (the reference filters image 0 of the batch, test.py:108-110)")
    curr_bs_l = IA_lab_large[:, 0:1, :, :].contiguous()
    curr_predict = upsample_ab(I_current_ab_predict.detach().contiguous().float())
    if tuple(curr_predict.shape[-2:]) != tuple(curr_bs_l.shape[-2:]):
        raise RuntimeError(f"frame_tail: upsampled ab {tuple(curr_predict.shape[-2:])} vs L {tuple(curr_bs_l.shape[-2:])}")
    if wls_filter_on:
        guide_image = luminance_guide_u8(curr_bs_l[0, 0])
        curr_predict = fgs_filter(guide_image, curr_predict[0], lambda_value, sigma_color).unsqueeze(0)
    IA_predict_rgb = lab_to_rgb8(curr_bs_l[0, 0], curr_predict[0])
    return IA_predict_rgb, curr_predict


def frames_tail(IA_lab_large_list, ab_list, wls_filter_on=True, lambda_value=500, sigma_color=4):
    """`frame_tail` for several frames at once (same results): the filter of G frames is one set of launches
    with G x more independent lines in flight, which is what this latency-bound filter lacks."""
    G = len(ab_list)
    L = torch.cat([f[:, 0:1] for f in IA_lab_large_list], dim=0).contiguous()          # [G,1,2H,2W]
    cur = upsample_ab(torch.cat([a.detach().float() for a in ab_list], dim=0).contiguous())   # [G,2,2H,2W]
    if wls_filter_on:
        cur = fgs_filter(luminance_guide_u8(L[:, 0]), cur, lambda_value, sigma_color)
    return [lab_to_rgb8(L[g, 0], cur[g]) for g in range(G)], cur
