"""CPU restatement of the whole clip driver, /root/reference/test.py:29-124 minus its file I/O:
the composition of the other oracle modules in the order test.py runs them.

TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product package).

    transform = Compose([CenterPad, CenterCrop, RGB2Lab, ToTensor, Normalize])        test.py:44-46   ingest_oracle.frame_ingest
    IB_lab    = F.interpolate(transform(frame_ref), 0.5, "bilinear")                  test.py:57-58   tail_oracle.downsample_half
    features_B = vggnet(tensor_lab2rgb(uncenter_l(IB_l), IB_ab))                      test.py:61-66   dvc_oracle.exemplar_features
    per frame: IA_lab = F.interpolate(transform(frame), 0.5); I_last = 0 | IB_lab     test.py:69-80
               ab = frame_colorization(IA_lab, IB_lab, I_last, ..., T = 1e-10)        test.py:83-95   dvc_oracle.frame_colorization
               I_last = cat(IA_l, ab)                                                 test.py:96
               x2 bilinear * 1.25 -> WLS filter -> Lab -> 8-bit RGB                   test.py:98-116  tail_oracle.frame_tail

Pinning status = that of the parts: the network path is pinned bit-exact against the reference modules
(oracle/pin_reference.py); the ingest's resize / rgb2lab and the tail's WLS filter / lab2rgb restate third-party code
(scikit-image, opencv-contrib) that is absent from this image — **parity unpinned** for those four stages
(ingest_oracle.py, tail_oracle.py say exactly what is restated).
"""
import numpy as np
import torch

from . import dvc_oracle, ingest_oracle, tail_oracle


def colorize_video(frames_rgb8, reference_rgb8, image_size, sd_vgg, sd_warp, sd_color, frame_propagate=False,
                   wls_filter_on=True, lambda_value=500, sigma_color=4, taps=None):
    """frames_rgb8: list of H0 x W0 x 3 uint8 arrays in the order test.py:41 sorts the files; reference_rgb8: the
    reference image (ignored with frame_propagate, where test.py:50 takes the first frame).  image_size = opt.image_size
    as test.py:163 leaves it (twice the network resolution).  Returns the list of H x W x 3 uint8 frames
    `save_frames` receives (test.py:120).  `taps` (dict, optional) receives the per-frame ab predictions and the
    smallest top-1/top-2 affinity gap of each frame's correlation."""
    large = [torch.from_numpy(ingest_oracle.frame_ingest(np.asarray(f), image_size))[None] for f in frames_rgb8]
    ref_large = large[0] if frame_propagate else \
        torch.from_numpy(ingest_oracle.frame_ingest(np.asarray(reference_rgb8), image_size))[None]
    half = lambda t: torch.from_numpy(tail_oracle.downsample_half(t.numpy()))           # noqa: E731  (test.py:58,71)
    IB_lab = half(ref_large)
    outs, abs_, gaps = [], [], []
    with torch.no_grad():
        features_B = dvc_oracle.exemplar_features(IB_lab, sd_vgg)
        I_last_lab_predict = None
        for IA_lab_large in large:
            IA_lab = half(IA_lab_large)
            if I_last_lab_predict is None:
                I_last_lab_predict = IB_lab if frame_propagate else torch.zeros_like(IA_lab)
            t = {} if taps is not None else None
            ab, _, _ = dvc_oracle.frame_colorization(IA_lab, IB_lab, I_last_lab_predict, features_B, sd_vgg, sd_warp,
                                                     sd_color, temperature=1e-10, taps=t)
            I_last_lab_predict = torch.cat((IA_lab[:, 0:1], ab), dim=1)
            rgb, _ = tail_oracle.frame_tail(IA_lab_large[:, 0:1].numpy(), ab.numpy(), wls_filter_on, float(lambda_value),
                                            float(sigma_color))
            outs.append(rgb)
            abs_.append(ab)
            if t is not None:
                gaps.append((t["top2"][0, :, 0] - t["top2"][0, :, 1]).min().item())
    if taps is not None:
        taps.update(ab=abs_, min_gap=gaps, IB_lab=IB_lab)
    return outs
