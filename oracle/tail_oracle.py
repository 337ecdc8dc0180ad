"""CPU restatement of the clip-driver tail, /root/reference/test.py:98-116.

TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/, never by the product package).

    curr_predict = F.interpolate(ab, scale_factor=2, mode="bilinear") * 1.25             test.py:100-102
    guide        = (uncenter_l(L_large) * 255 / 100).astype(uint8)                       test.py:106-109
    a', b'       = cv2.ximgproc FastGlobalSmootherFilter(guide, lambda=500, sigma=4)      test.py:107-111
    rgb          = batch_lab2rgb_transpose_mc(L_large, (a', b'))  (skimage lab2rgb)       utils/util.py:134-151

Pinning status:
  * `upsample_ab` is pinned: the reference calls torch.nn.functional.interpolate itself, and
    tests/test_tail.py checks this restatement against it bit for bit.
  * `luminance_guide_u8` is plain arithmetic (float32 multiply / divide / truncation), as numpy does it.
  * `fgs_filter` and `lab_to_rgb8` are **parity unpinned**: cv2.ximgproc (opencv-contrib) and skimage are
    third-party dependencies that are neither vendored in /root/reference nor installed in this image
    (requirements.txt lists `opencv-contrib-python`, `scikit-image`, unpinned).  `fgs_filter` restates the
    published algorithm the OpenCV filter implements — D. Min et al., "Fast Global Image Smoothing Based on
    Weighted Least Squares", IEEE TIP 2014, Algorithm 1: T = 3 iterations of separable 1-D WLS solves
    (horizontal then vertical) with lambda_t = 1.5 * 4^(T-t) / (4^T - 1) * lambda and range weights
    exp(-|g_p - g_q| / sigma_color) on the 8-bit guide; `lab_to_rgb8` restates skimage.color.lab2rgb
    (D65, 2 degree observer, float64) followed by clip / *255 / astype(uint8) as utils/util.py:134 does.
"""
import numpy as np


def upsample_ab(ab):
    """F.interpolate(ab, scale_factor=2, mode='bilinear', align_corners=False) * 1.25; ab: [N,C,H,W] float32.

    ATen (UpSampleKernel.cpp, separable weights): source index = max(0, 0.5*(dst+0.5) - 0.5), i0 = floor,
    i1 = min(i0+1, S-1), w1 = src - i0, w0 = 1 - w1;  out = wy0*(wx0*x00 + wx1*x01) + wy1*(wx0*x10 + wx1*x11)."""
    ab = np.asarray(ab, dtype=np.float32)
    N, C, H, W = ab.shape

    def axis(S):
        d = np.arange(2 * S, dtype=np.float32)
        src = np.maximum(np.float32(0.5) * (d + np.float32(0.5)) - np.float32(0.5), np.float32(0))
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, S - 1)
        w1 = (src - i0.astype(np.float32)).astype(np.float32)
        w0 = (np.float32(1) - w1).astype(np.float32)
        return i0, i1, w0, w1

    def fma(w0, a, w1, b):
        # ATen's compiled kernel evaluates w0*a + w1*b as fma(w0, a, fl32(w1*b)) (found by matching its output
        # bit for bit); float32 products are exact in float64, so this emulation is exact up to a double
        # rounding that the test below has never seen
        t = (w1 * b).astype(np.float32)
        return (w0.astype(np.float64) * a.astype(np.float64) + t.astype(np.float64)).astype(np.float32)

    y0, y1, wy0, wy1 = axis(H)
    x0, x1, wx0, wx1 = axis(W)
    r0, r1 = ab[:, :, y0], ab[:, :, y1]
    shape = r0[:, :, :, x0].shape
    WX0, WX1 = np.broadcast_to(wx0, shape), np.broadcast_to(wx1, shape)
    WY0, WY1 = np.broadcast_to(wy0[:, None], shape), np.broadcast_to(wy1[:, None], shape)
    top = fma(WX0, r0[:, :, :, x0], WX1, r0[:, :, :, x1])
    bot = fma(WX0, r1[:, :, :, x0], WX1, r1[:, :, :, x1])
    out = fma(WY0, top, WY1, bot)
    return (out * np.float32(1.25)).astype(np.float32)


def downsample_half(x):
    """F.interpolate(x, scale_factor=0.5, mode='bilinear') (test.py:58,71).  ATen evaluates it exactly like
    F.avg_pool2d(x, 2): ((x00 + x01) + x10) + x11, then * 0.25 (pinned against both calls in tests/test_tail.py)."""
    x = np.asarray(x, dtype=np.float32)
    H2, W2 = x.shape[2] // 2, x.shape[3] // 2
    x = x[:, :, :2 * H2, :2 * W2]
    s = (x[:, :, 0::2, 0::2] + x[:, :, 0::2, 1::2]).astype(np.float32)
    s = (s + x[:, :, 1::2, 0::2]).astype(np.float32)
    s = (s + x[:, :, 1::2, 1::2]).astype(np.float32)
    return (s * np.float32(0.25)).astype(np.float32)


def luminance_guide_u8(L_centered):
    """(uncenter_l(L) * 255 / 100).astype(uint8): test.py:106-109, utils/util.py:63-64."""
    L = np.asarray(L_centered, dtype=np.float32)
    g = (L + np.float32(50.0)) * np.float32(255.0) / np.float32(100.0)
    return g.astype(np.uint8)


def _solve_rows(f, w, lam):
    """One separable WLS pass along axis 1: (I + lam*A) u = f per row, A = graph Laplacian of the chain with
    edge weights w[:, x] between pixels x and x+1 (w has W-1 useful columns).  Thomas algorithm, float32."""
    H, W = f.shape
    a = np.zeros((H, W), np.float32)
    c = np.zeros((H, W), np.float32)
    a[:, 1:] = -lam * w[:, :W - 1]
    c[:, :W - 1] = -lam * w[:, :W - 1]
    b = (np.float32(1) - a - c).astype(np.float32)
    cp = np.zeros((H, W), np.float32)
    dp = np.zeros((H, W), np.float32)
    cp[:, 0] = c[:, 0] / b[:, 0]
    dp[:, 0] = f[:, 0] / b[:, 0]
    for x in range(1, W):
        m = (b[:, x] - a[:, x] * cp[:, x - 1]).astype(np.float32)
        cp[:, x] = c[:, x] / m
        dp[:, x] = (f[:, x] - a[:, x] * dp[:, x - 1]) / m
    u = np.zeros((H, W), np.float32)
    u[:, W - 1] = dp[:, W - 1]
    for x in range(W - 2, -1, -1):
        u[:, x] = dp[:, x] - cp[:, x] * u[:, x + 1]
    return u


def fgs_filter(guide_u8, src, lambda_value=500.0, sigma_color=4.0, num_iter=3, lambda_attenuation=0.25):
    """Fast global smoother (Min et al. 2014, Alg. 1) of one float32 plane `src` [H,W] with an 8-bit guide."""
    g = np.asarray(guide_u8).astype(np.int32)
    lut = np.exp(-np.arange(256, dtype=np.float32) / np.float32(sigma_color)).astype(np.float32)
    wh = lut[np.abs(g[:, 1:] - g[:, :-1])]          # [H, W-1]  weight between x and x+1
    wv = lut[np.abs(g[1:, :] - g[:-1, :])]          # [H-1, W]  weight between y and y+1
    u = np.asarray(src, dtype=np.float32).copy()
    lam = np.float32(1.5 * lambda_value * 4.0 ** (num_iter - 1) / (4.0 ** num_iter - 1.0))
    for _ in range(num_iter):
        u = _solve_rows(u, wh, lam)
        u = _solve_rows(u.T.copy(), wv.T.copy(), lam).T.copy()
        lam = np.float32(lam * np.float32(lambda_attenuation))
    return u


_XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423],
                          [0.212671, 0.715160, 0.072169],
                          [0.019334, 0.119193, 0.950227]], dtype=np.float64)
RGB_FROM_XYZ = np.linalg.inv(_XYZ_FROM_RGB)
_D65 = np.array([0.95047, 1.0, 1.08883], dtype=np.float64)


def lab_to_rgb8(L_centered, ab):
    """utils/util.py:134-151 for one image: Lab (L + 50 in [0,100]) -> skimage.color.lab2rgb (float64) ->
    clip [0,1] * 255 -> uint8, HWC.  L_centered: [H,W], ab: [2,H,W]."""
    L = np.asarray(L_centered, dtype=np.float32).astype(np.float64) + 50.0
    a = np.asarray(ab[0], dtype=np.float32).astype(np.float64)
    b = np.asarray(ab[1], dtype=np.float32).astype(np.float64)
    fy = (L + 16.0) / 116.0
    fx = a / 500.0 + fy
    fz = fy - b / 200.0
    fz = np.maximum(fz, 0.0)
    xyz = np.stack([fx, fy, fz], axis=-1)
    big = xyz > 0.2068966
    xyz = np.where(big, xyz ** 3, (xyz - 16.0 / 116.0) / 7.787)
    xyz = xyz * _D65
    rgb = xyz @ RGB_FROM_XYZ.T
    hi = rgb > 0.0031308
    rgb = np.where(hi, 1.055 * np.power(np.maximum(rgb, 1e-300), 1.0 / 2.4) - 0.055, rgb * 12.92)
    return (np.clip(rgb, 0.0, 1.0) * 255.0).astype(np.uint8)


def rgb8_to_lab(rgb_hwc):
    """Normalize()(ToTensor()(RGB2Lab()(image))): skimage.color.rgb2lab (float64) of a uint8 H x W x 3 image,
    .float(), L - 50 (utils/util_distortion.py:18-23,85-100, lib/functional.py:85-103).  Returns [3,H,W] float32.
    Parity unpinned (skimage absent): the published sRGB -> XYZ (D65) -> CIELAB formulas skimage implements."""
    c = np.asarray(rgb_hwc, dtype=np.uint8).astype(np.float64) / 255.0
    c = np.where(c > 0.04045, np.power((c + 0.055) / 1.055, 2.4), c / 12.92)
    xyz = c @ _XYZ_FROM_RGB.T
    xyz = xyz / _D65
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    L = (116.0 * f[..., 1] - 16.0).astype(np.float32) - np.float32(50.0)
    a = (500.0 * (f[..., 0] - f[..., 1])).astype(np.float32)
    b = (200.0 * (f[..., 1] - f[..., 2])).astype(np.float32)
    return np.stack([L, a, b]).astype(np.float32)


def frame_tail(L_large_centered, ab_predict, wls_filter_on=True, lambda_value=500.0, sigma_color=4.0):
    """test.py:98-116 for batch 1.  L_large_centered: [1,1,2H,2W], ab_predict: [1,2,H,W].
    Returns (rgb uint8 [2H,2W,3], filtered ab float32 [1,2,2H,2W])."""
    cur = upsample_ab(ab_predict)
    if wls_filter_on:
        guide = luminance_guide_u8(np.asarray(L_large_centered)[0, 0])
        cur = np.stack([fgs_filter(guide, cur[0, 0], lambda_value, sigma_color),
                        fgs_filter(guide, cur[0, 1], lambda_value, sigma_color)])[None]
    rgb = lab_to_rgb8(np.asarray(L_large_centered)[0, 0], cur[0])
    return rgb, cur
