"""Tensor helpers on the hot path, HIP-backed (mirror of the ★ rows of /root/reference/utils/util.py).

Only the functions the inference path uses are provided; host I/O, plotting and
training-loss helpers of the reference's util.py are out of scope.
"""
import torch

from . import ops

# l: [-50,50]; ab: [-128,128]   (utils/util.py:15-18)
l_norm, ab_norm = 1.0, 1.0
l_mean, ab_mean = 50.0, 0


def center_l(l):
    """utils/util.py:56-59"""
    return (l - l_mean) / l_norm


def uncenter_l(l):
    """utils/util.py:63-64 (plain arithmetic: works on floats, arrays and tensors alike)."""
    return l * l_norm + l_mean


def center_ab(ab):
    """utils/util.py:68-69"""
    return (ab - ab_mean) / ab_norm


def gray2rgb_batch(l):
    """utils/util.py:97-101 — gray (centred L) tensor to a 3-channel [0,1] tensor."""
    return ops.gray2rgb(l)


def feature_normalize(feature_in):
    """utils/util.py:155-158 — divide by the channel L2 norm (+ float64 epsilon)."""
    if not feature_in.is_cuda:
        raise RuntimeError("feature_normalize: the HIP path has no CPU fallback")
    return ops.channel_l2norm(feature_in.detach().contiguous().float())


def vgg_preprocess(tensor):
    """utils/util.py:347-352 — RGB [0,1] -> BGR, minus Caffe mean, x255.

    VGG19_pytorch.forward(preprocess=True) folds this into conv1_1's load; this standalone version
    exists for API completeness and runs as a 1x1 'conv' on the same engine."""
    if not tensor.is_cuda:
        raise RuntimeError("vgg_preprocess: the HIP path has no CPU fallback")
    x = tensor.detach().contiguous().float()
    N = x.shape[0]
    dev = x.device
    # identity-with-channel-swap weights [Cin=3][1][Cout=4] (Cout padded to 4 for the engine)
    w = torch.zeros(3, 1, 4, device=dev)
    w[2, 0, 0] = 1.0
    w[1, 0, 1] = 1.0
    w[0, 0, 2] = 1.0
    mean_rgb = torch.tensor([0.48501961, 0.45795686, 0.40760392], device=dev)
    sc = torch.full((3,), 255.0, device=dev).repeat(N)
    sh = (-255.0 * mean_rgb).repeat(N)
    y = ops.conv2d(x, w, None, ksize=1, pad=0, in_scale=sc, in_shift=sh)
    return y[:, 0:3].contiguous()


def tensor_lab2rgb(input):
    """utils/util.py:379-414 — n x 3 x h x w Lab (L in [0,100]) -> sRGB [0,1]."""
    if not input.is_cuda:
        raise RuntimeError("tensor_lab2rgb: the HIP path has no CPU fallback")
    return ops.lab2rgb(input.detach().contiguous().float())
