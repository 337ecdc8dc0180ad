"""CPU restatement of the frame ingest, /root/reference/test.py:44-46,69-71.

TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/, never by the product package).

    transform = Compose([CenterPad(image_size), CenterCrop(image_size), RGB2Lab(), ToTensor(), Normalize()])
    IA_lab_large = transform(frame).unsqueeze(0).cuda()

`center_pad` follows utils/util_distortion.py:217-258 line by line.  Its `skimage.transform.resize(I, new_size,
mode="reflect", preserve_range=True, clip=False, anti_aliasing=True)` is **parity unpinned**: scikit-image is a
third-party dependency that is neither vendored in /root/reference nor installed here (requirements.txt lists
`scikit-image`, unpinned).  What is restated is what scikit-image >= 0.19 documents and does for that call —
and it does it with SciPy, which IS installed, so the arithmetic below is SciPy's own:
    factors = input_shape / output_shape;   sigma = max(0, (factors - 1) / 2)            (anti-aliasing)
    filtered = scipy.ndimage.gaussian_filter(image, sigma, mode="mirror", cval=0)          (skimage "reflect" ==
                                                                                            SciPy "mirror")
    out = scipy.ndimage.zoom(filtered, 1 / factors, order=1, mode="mirror", grid_mode=True)
(`resize_numpy` spells the same two steps out index by index; tests/test_ingest.py checks it against the SciPy
calls, and the HIP kernels against both.)  The colour half (RGB2Lab -> ToTensor -> Normalize) is
tail_oracle.rgb8_to_lab.
"""
import numpy as np
import scipy.ndimage as ndi

from . import tail_oracle


def skimage_resize(I, new_size):
    """resize(I, new_size, mode="reflect", preserve_range=True, clip=False, anti_aliasing=True) for an H x W x C
    array (any real dtype) -> float64 [new_h, new_w, C]."""
    I = np.asarray(I).astype(np.float64)
    factors = np.array(I.shape[:2], dtype=np.float64) / np.array(new_size, dtype=np.float64)
    sigma = np.maximum(0.0, (factors - 1.0) / 2.0)
    if (sigma > 0).any():
        I = ndi.gaussian_filter(I, (sigma[0], sigma[1], 0.0), cval=0.0, mode="mirror")
    return ndi.zoom(I, (1.0 / factors[0], 1.0 / factors[1], 1.0), order=1, mode="mirror", cval=0.0, grid_mode=True)


def _mirror(i, n):
    if n == 1:
        return np.zeros_like(i)
    p = 2 * n - 2
    m = np.mod(i, p)
    return np.where(m >= n, p - m, m)


def gaussian_weights(sigma):
    """SciPy's gaussian_filter1d kernel: radius int(4 sigma + 0.5), exp(-x^2 / (2 sigma^2)), normalised."""
    r = int(4.0 * sigma + 0.5)
    k = np.exp(-0.5 * (np.arange(-r, r + 1, dtype=np.float64) / sigma) ** 2)
    return k / k.sum()


def resize_numpy(I, new_size):
    """`skimage_resize`, index by index (what csrc/ingest.hip implements)."""
    x = np.asarray(I).astype(np.float64)
    for axis in (0, 1):
        n = x.shape[axis]
        factor = n / float(new_size[axis])
        sigma = max(0.0, (factor - 1.0) / 2.0)
        if sigma > 0:
            k = gaussian_weights(sigma)
            r = (len(k) - 1) // 2
            acc = np.zeros_like(x)
            for j in range(-r, r + 1):
                acc += k[j + r] * np.take(x, _mirror(np.arange(n) + j, n), axis=axis)
            x = acc
    for axis in (0, 1):
        n, n_out = x.shape[axis], int(new_size[axis])
        cc = (np.arange(n_out, dtype=np.float64) + 0.5) * (n / float(n_out)) - 0.5
        i0 = np.floor(cc).astype(np.int64)
        t = cc - i0
        shape = [1] * x.ndim
        shape[axis] = n_out
        t = t.reshape(shape)
        x = np.take(x, _mirror(i0, n), axis=axis) * (1.0 - t) + np.take(x, _mirror(i0 + 1, n), axis=axis) * t
    return x


def center_pad(image_u8, image_size, resize=skimage_resize):
    """CenterPad(image_size)(image) of utils/util_distortion.py:217-258; image_u8: H0 x W0 x 3 uint8."""
    I = np.asarray(image_u8)
    height_old, width_old = I.shape[0], I.shape[1]
    old_size = [height_old, width_old]
    height, width = int(image_size[0]), int(image_size[1])
    I_pad = np.zeros((height, width, I.shape[2]))
    ratio = height / width
    if height_old / width_old == ratio:
        if height_old == height:
            return I.astype(np.uint8)
        new_size = [int(x * height / height_old) for x in old_size]
        return resize(I, new_size).astype(np.uint8)
    if height_old / width_old > ratio:      # resize to the target width, crop the height
        new_size = [int(x * width / width_old) for x in old_size]
        I_resize = resize(I, new_size)
        start_height = (I_resize.shape[0] - height) // 2
        I_pad[:, :, :] = I_resize[start_height:(start_height + height), :, :]
    else:                                    # resize to the target height, crop the width
        new_size = [int(x * height / height_old) for x in old_size]
        I_resize = resize(I, new_size)
        start_width = (I_resize.shape[1] - width) // 2
        I_pad[:, :, :] = I_resize[:, start_width:(start_width + width), :]
    return I_pad.astype(np.uint8)


def frame_ingest(image_u8, image_size):
    """transform(frame) of test.py:44-46 (CenterCrop(image_size) is the identity after CenterPad): uint8
    H0 x W0 x 3 -> centred Lab float32 [3, H, W]."""
    return tail_oracle.rgb8_to_lab(center_pad(image_u8, image_size))
