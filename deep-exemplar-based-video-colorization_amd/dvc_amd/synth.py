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


def colorvidnet_state_dict(seed=0, ic=7, contractive=False):
    """`contractive=False`: plain He-uniform weights.  A 31-conv / 9-InstanceNorm stack with such weights is
    chaotic: it amplifies a relative perturbation ~70x (measured, fp64) and its own output ~40-100x per
    recurrence step, so the reference's CPU fp32 run differs from ITSELF by 6e-3 / 0.26 / 15.8 max-abs on
    free-running frames 0 / 1 / 2 when only the thread count changes.  Fine for per-stage tests, useless
    for an end-to-end tolerance.

    `contractive=True`: every 3x3 convolution is identity-on-the-centre-tap + 0.3 x the same He-uniform
    draw (channel c of the output follows channel c mod Cin of the input; wide->narrow convolutions average
    the folded channels), and the output head is scaled by CONTRACTIVE_OUT_GAIN.  Every channel stays
    active and 40 % of each layer's output energy is still the random mixture (a wrong layer changes the
    result at O(1)), but relative perturbations no longer grow through the stack (fp32-vs-fp64 stays ~1e-6
    relative at every tap) and the recurrence gain is below one: the reference-equivalent CPU fp32 run agrees
    with itself across thread counts to < 1e-4 on every frame of a free-running clip (tests/
    test_gpu_e2e.py states the measured numbers), which is what makes the north-star tolerance
    (ab within 1e-3 max-abs) assertable literally."""
    sd = _fill(arch.colorvidnet_param_shapes(ic), seed + 2, math.sqrt(2.0))
    # keep tanh(conv10_ab) away from saturation so the ab output is informative
    sd["conv10_ab.weight"] = sd["conv10_ab.weight"] * 0.25
    if contractive:
        for k, v in list(sd.items()):
            if k.endswith(".weight") and v.dim() == 4 and v.shape[2] == 3:
                co, ci = v.shape[:2]
                eye = torch.zeros_like(v)
                eye[torch.arange(co), torch.arange(co) % ci, 1, 1] = 1.0
                if co < ci:
                    eye[torch.arange(ci) % co, torch.arange(ci), 1, 1] = 1.0
                    eye = eye / math.sqrt(ci / co)
                sd[k] = eye + CONTRACTIVE_MIX * v
        sd["conv10_ab.weight"] = sd["conv10_ab.weight"] * CONTRACTIVE_OUT_GAIN
        sd["conv10_ab.bias"] = sd["conv10_ab.bias"] * CONTRACTIVE_OUT_GAIN
    return sd


CONTRACTIVE_MIX = 0.3
CONTRACTIVE_OUT_GAIN = 0.04
# Frames for the literal end-to-end tolerance at test.py's temperature (1e-10 = hard arg-max): seeds of
# FRAME_SEED0.. whose smallest top-1/top-2 affinity gap over all 5184 query rows is >= 2e-6 against
# exemplar seed 2 at 216x384 with the seed-0 VGG/WarpNet weights (measured with the oracle; 8 of the 12 seeds
# 1000..1011 have a row below 2e-6, where fp32 rounding — ~3e-7 on an affinity — decides the arg-max and
# the reference itself flips with its thread count).
WELL_SEPARATED_FRAME_SEEDS_216x384 = (1000, 1003, 1006, 1009)
# same criterion at 432x768 (20736 rows): seed 1001 (smallest gap 2.7e-6; 13 of the 14 seeds 1000..1013 have a row
# below 2e-6)
WELL_SEPARATED_FRAME_SEED_432x768 = 1001


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
