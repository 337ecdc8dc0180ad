"""Deterministic synthetic weights and inputs (there is no network for checkpoints/datasets).

The reference's released weights (`data/vgg19_conv.pth`, `checkpoints/video_moredata_l1/*.pth`,
/root/reference/test.py:150-159) are not in the tree, so every parity test and the benchmark use
weights produced here.  Each tensor is drawn from its own `torch.Generator` seeded from
(seed, crc32(key)), so a state_dict is reproducible tensor-by-tensor on any machine with the same
torch CPU generator, independent of construction order.  He-style uniform bounds keep activations
O(1) through the ReLU stacks so that comparisons are not dominated by vanishing signals.

`synth_lab` is the smooth Lab-frame generator specified in SURVEY.md §8(d).
"""
import math
import zlib

import torch
import torch.nn.functional as F

from . import arch


def _gen(seed, key):
    g = torch.Generator()
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    return g


def _fill(shapes, seed, gain):
    sd = {}
    for key, shape in shapes.items():
        g = _gen(seed, key)
        if len(shape) == 4 and shape[1:] == (1, 1, 1):          # depthwise 1x1 "subsample" scale
            t = 0.5 + torch.rand(shape, generator=g)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            bound = gain * math.sqrt(3.0 / fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif shape == (1,):                                      # PReLU slope
            t = 0.1 + 0.3 * torch.rand(shape, generator=g)
        else:                                                    # conv bias
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
        sd[key] = t.float()
    return sd


def vgg19_state_dict(seed=0):
    return _fill(arch.vgg_param_shapes(), seed, math.sqrt(2.0))


def warpnet_state_dict(seed=0):
    return _fill(arch.warpnet_param_shapes(), seed + 1, math.sqrt(2.0))


def colorvidnet_state_dict(seed=0, ic=7):
    sd = _fill(arch.colorvidnet_param_shapes(ic), seed + 2, math.sqrt(2.0))
    # keep tanh(conv10_ab) away from saturation so the ab output is informative
    sd["conv10_ab.weight"] = sd["conv10_ab.weight"] * 0.25
    return sd


def synth_lab(seed, H=216, W=384):
    """Smooth synthetic Lab frame, 1x3xHxW fp32: L centred in [-50, 55], ab in [-80, 88]."""
    g = torch.Generator()
    g.manual_seed(int(seed))
    base = torch.rand(1, 3, max(H // 8, 1), max(W // 8, 1), generator=g)
    x = F.interpolate(base, (H, W), mode="bilinear", align_corners=False)
    x = x + 0.05 * torch.rand(1, 3, H, W, generator=g)
    return torch.cat((x[:, 0:1] * 100 - 50, (x[:, 1:3] - 0.5) * 160), dim=1).contiguous()


EXEMPLAR_SEED = 2
FRAME_SEED0 = 1000
