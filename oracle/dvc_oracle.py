"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A functional (state_dict-driven) restatement, on the PyTorch CPU backend, of the reference's
inference hot path: VGG19 features -> WarpNet dense correlation -> ColorVidNet generator.  It is the
checker that the HIP path is compared against; it is never the thing that is shipped or measured
(only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it).

Why PyTorch CPU and not numpy/C: the reference has no native code; all of its arithmetic *is*
ATen's CPU kernels reached through `torch.nn` ("Third-party arithmetic"), torch is
unpinned in /root/reference/requirements.txt:11, and this image's torch 2.10.0 is the build the
reference modules run on here.  Each function below keeps the reference's op order so that, given
the same state_dict and inputs, it reproduces the reference module bit-for-bit on this backend.

Parity pinning: the reference ships no tests, golden vectors or weights for this path.  The oracle
is pinned instead by `oracle/pin_reference.py`, which imports the *unmodified* reference modules
from /root/reference (in the build container), loads the same synthetic state_dicts, and checks
this restatement against them (bit-exact on every tap); the outputs it saves to `tests/golden/`
are what `tests/test_oracle_golden.py` re-checks wherever /root/reference is absent.

All functions accept fp32 or fp64 tensors (the fp64 run is the "truth" both fp32 implementations
are scored against, SURVEY.md §7 hard part 1).
"""
import sys

import torch
import torch.nn.functional as F

EPS = sys.float_info.epsilon  # utils/util.py:156, models/NonlocalNet.py:470,475


# ------------------------------------------------------------------------------- tensor helpers
def uncenter_l(l):
    """utils/util.py:63-64 (l_norm=1, l_mean=50)."""
    return l * 1.0 + 50.0


def gray2rgb_batch(l):
    """utils/util.py:97-101: (L+50)/100 replicated to three channels."""
    v = uncenter_l(l) / (2 * 50.0)
    return torch.cat((v, v, v), dim=1)


def vgg_preprocess(x):
    """utils/util.py:347-352: RGB[0,1] -> BGR, minus Caffe mean, times 255."""
    bgr = torch.cat((x[:, 2:3], x[:, 1:2], x[:, 0:1]), dim=1)
    mean = torch.Tensor([0.40760392, 0.45795686, 0.48501961]).type_as(bgr).view(1, 3, 1, 1)
    return (bgr - mean) * 255


def feature_normalize(x):
    """utils/util.py:155-158: divide by the channel L2 norm (+ float64 epsilon)."""
    return torch.div(x, torch.norm(x, 2, 1, keepdim=True) + EPS)


_XYZ2RGB = [[3.24048134, -0.96925495, 0.05564664],
            [-1.53715152, 1.87599, -0.20404134],
            [-0.49853633, 0.04155593, 1.05731107]]


def tensor_lab2rgb(lab):
    """utils/util.py:379-414.  Lab (L in [0,100]) -> sRGB in [0,1]; n x 3 x h x w.

    The two transcendental branches are evaluated on the boolean-gathered subsets, as the reference
    does: ATen's vectorised pow() and its scalar tail differ in the last ulp, so evaluating on the
    full tensor instead would move a handful of pixels by 1 ulp — which VGG's x255 preprocessing
    and the argmax downstream amplify to ~4e-3 on the ab output.
    """
    t = lab.permute(0, 2, 3, 1)
    L, a, b = t[..., 0:1], t[..., 1:2], t[..., 2:3]
    y = (L + 16.0) / 116.0
    x = (a / 500.0) + y
    z = (y - (b / 200.0)).clamp(min=0)
    xyz = torch.cat((x, y, z), dim=3)
    cube = xyz > 0.2068966
    lin = torch.empty_like(xyz)
    lin[cube] = torch.pow(xyz[cube], 3.0)
    lin[~cube] = (xyz[~cube] - 16.0 / 116.0) / 7.787
    white = torch.tensor([0.95047, 1.0, 1.08883], dtype=lin.dtype)
    lin = lin * white
    m = torch.tensor(_XYZ2RGB, dtype=torch.float64).type_as(lin)
    rgb = torch.mm(lin.reshape(-1, 3), m).view(lab.size(0), lab.size(2), lab.size(3), 3)
    rgb = rgb.permute(0, 3, 1, 2)
    gam = rgb > 0.0031308
    out = torch.empty_like(rgb)
    out[gam] = 1.055 * torch.pow(rgb[gam], 1 / 2.4) - 0.055
    out[~gam] = rgb[~gam] * 12.92
    return out.clamp(0, 1)


# ---------------------------------------------------------------------------------------- VGG19
_VGG_SEQ = [  # (key, conv name or None for pool)  models/NonlocalNet.py:235-255
    ("r11", "conv1_1"), ("r12", "conv1_2"), ("p1", None),
    ("r21", "conv2_1"), ("r22", "conv2_2"), ("p2", None),
    ("r31", "conv3_1"), ("r32", "conv3_2"), ("r33", "conv3_3"), ("r34", "conv3_4"), ("p3", None),
    ("r41", "conv4_1"), ("r42", "conv4_2"), ("r43", "conv4_3"), ("r44", "conv4_4"), ("p4", None),
    ("r51", "conv5_1"), ("r52", "conv5_2"), ("r53", "conv5_3"), ("r54", "conv5_4"), ("p5", None),
]


def vgg19_forward(sd, x, out_keys, preprocess=True, pool="max"):
    """VGG19_pytorch.forward, models/NonlocalNet.py:228-256.  Returns a list in out_keys order."""
    out = {}
    if preprocess:
        x = vgg_preprocess(x)
    cur = x
    for key, conv in _VGG_SEQ:
        if conv is None:
            cur = F.max_pool2d(cur, 2, 2) if pool == "max" else F.avg_pool2d(cur, 2, 2)
        else:
            cur = F.relu(F.conv2d(cur, sd[conv + ".weight"], sd[conv + ".bias"], padding=1))
        out[key] = cur
    return [out[k] for k in out_keys]


# -------------------------------------------------------------------------------------- WarpNet
def _inorm(x):
    """nn.InstanceNorm2d defaults: eps 1e-5, no affine, biased variance, input statistics."""
    return F.instance_norm(x, None, None, None, None, True, 0.1, 1e-5)


def _rconv(x, w, b, stride=1):
    """ReflectionPad2d(1) + Conv2d(k=3, padding=0)."""
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b, stride=stride)


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="nearest")


def warp_head(sd, name, x):
    """One of WarpNet.layer2_1 .. layer5_1, models/NonlocalNet.py:364-410."""
    if name == "layer5_1":
        ia, pa, ib, pb = 1, 3, 6, 8
    else:
        ia, pa, ib, pb = 1, 3, 5, 7
    stride_b = 2 if name == "layer2_1" else 1
    x = _rconv(x, sd[f"{name}.{ia}.weight"], sd[f"{name}.{ia}.bias"])
    x = F.prelu(_inorm(x), sd[f"{name}.{pa}.weight"])
    if name == "layer5_1":
        x = _up2(x)
    x = _rconv(x, sd[f"{name}.{ib}.weight"], sd[f"{name}.{ib}.bias"], stride=stride_b)
    x = F.prelu(_inorm(x), sd[f"{name}.{pb}.weight"])
    if name in ("layer4_1", "layer5_1"):
        x = _up2(x)
    return x


def residual_block(sd, prefix, x):
    """ResidualBlock.forward, models/NonlocalNet.py:341-352 (one PReLU shared by both sites)."""
    a = sd[prefix + ".prelu.weight"]
    out = _rconv(x, sd[prefix + ".conv1.weight"], sd[prefix + ".conv1.bias"])
    out = F.prelu(_inorm(out), a)
    out = _rconv(out, sd[prefix + ".conv2.weight"], sd[prefix + ".conv2.bias"])
    out = _inorm(out)
    out = out + x
    return F.prelu(out, a)


def warp_features(sd, r2, r3, r4, r5):
    """Heads + concat + 3 residual blocks for ONE side (A or B), NonlocalNet.py:451-465."""
    f2 = warp_head(sd, "layer2_1", r2)
    f3 = warp_head(sd, "layer3_1", r3)
    f4 = warp_head(sd, "layer4_1", r4)
    f5 = warp_head(sd, "layer5_1", r5)
    if f5.shape[2] != f2.shape[2] or f5.shape[3] != f2.shape[3]:
        f5 = F.pad(f5, (0, 0, 1, 1), "replicate")
    x = torch.cat((f2, f3, f4, f5), 1)
    for b in range(3):
        x = residual_block(sd, f"layer.{b}", x)
    return x


class _WTAScale(torch.autograd.Function):
    """WTA_scale, models/NonlocalNet.py:288-327: forward keeps the row maximum and scales every other affinity by `scale`;
    backward multiplies the incoming gradient by 1 at the row maximum and by the CONSTANT 1e-4 elsewhere — whatever `scale`
    is (NonlocalNet.py:320-325) — which is not the derivative of the forward unless scale == 1e-4, and is restated as written."""

    @staticmethod
    def forward(ctx, f, scale):
        mx = torch.max(f, -1, keepdim=True)[0]
        mask = f == mx
        ctx.save_for_backward(mask)
        return torch.where(mask, f, f * scale)

    @staticmethod
    def backward(ctx, grad_output):
        (mask,) = ctx.saved_tensors
        return grad_output * torch.where(mask, torch.ones_like(grad_output), torch.full_like(grad_output, 1e-4)), None


def wta_scale(f, scale):
    """WTA_scale.apply(f, scale), models/NonlocalNet.py:288-327 (forward values as before r05; backward as the reference's)."""
    return _WTAScale.apply(f, scale)


def corr_project(sd, which, feats):
    """1x1 conv + centre over positions + L2-normalise over channels, NonlocalNet.py:468-476."""
    n = feats.shape[0]
    t = F.conv2d(feats, sd[which + ".weight"], sd[which + ".bias"]).view(n, 256, -1)
    t = t - t.mean(dim=-1, keepdim=True)
    t_norm = torch.norm(t, 2, 1, keepdim=True) + EPS
    return torch.div(t, t_norm)


def correlate(theta, phi, B_lab_map, temperature, WTA_scale_weight=1):
    """Affinity, similarity map, softmax, colour gather: NonlocalNet.py:477-500.

    theta, phi: n x 256 x N (centred + normalised).  Returns (y n x 3 x h x w at 1/4 res,
    similarity n x 1 x h x w at 1/4 res, f n x N x N).
    """
    n, channel, H, W = B_lab_map.shape
    fh, fw = int(H / 4), int(W / 4)
    f = torch.matmul(theta.permute(0, 2, 1), phi)
    sim = torch.max(f.unsqueeze(1), -1, keepdim=True)[0].view(n, 1, fh, fw)
    f_wta = f if WTA_scale_weight == 1 else wta_scale(f, WTA_scale_weight)
    f_wta = f_wta / temperature
    p = F.softmax(f_wta, dim=-1)
    B_lab = F.avg_pool2d(B_lab_map, 4).view(n, channel, -1).permute(0, 2, 1)
    y = torch.matmul(p, B_lab).permute(0, 2, 1).contiguous().view(n, channel, fh, fw)
    return y, sim, f


def correlate_chunked(theta, phi, B_lab_map, temperature, rows=2048):
    """`correlate` evaluated `rows` query rows at a time, so that the N x N matrix (1.72 GB fp32 at
    432x768, several copies of it inside `correlate`) never has to exist at once.  Same operations per
    row (matmul row block, max, / T, softmax, matmul with the pooled colours); a BLAS row block may round
    differently from the full product in the last place, so this is the oracle for sizes where
    `correlate` does not fit, and is itself checked against `correlate` at small sizes
    (tests/test_oracle_golden.py).  WTA_scale_weight == 1 only.
    Returns (y n x 3 x h x w, sim n x 1 x h x w, argmax n x N, top-1/top-2 gap n x N)."""
    n, channel, H, W = B_lab_map.shape
    fh, fw = int(H / 4), int(W / 4)
    N = theta.shape[2]
    B_lab = F.avg_pool2d(B_lab_map, 4).view(n, channel, -1).permute(0, 2, 1)
    tp = theta.permute(0, 2, 1)
    y = torch.empty(n, N, channel, dtype=theta.dtype)
    sim = torch.empty(n, N, dtype=theta.dtype)
    amax = torch.empty(n, N, dtype=torch.long)
    gap = torch.empty(n, N, dtype=theta.dtype)
    for r0 in range(0, N, rows):
        f = torch.matmul(tp[:, r0:r0 + rows], phi)
        top2 = torch.topk(f, 2, dim=-1)
        sim[:, r0:r0 + rows] = top2[0][..., 0]
        amax[:, r0:r0 + rows] = f.argmax(-1)
        gap[:, r0:r0 + rows] = top2[0][..., 0] - top2[0][..., 1]
        y[:, r0:r0 + rows] = torch.matmul(F.softmax(f / temperature, dim=-1), B_lab)
    y = y.permute(0, 2, 1).contiguous().view(n, channel, fh, fw)
    return y, sim.view(n, 1, fh, fw), amax, gap


def warpnet_forward(sd, B_lab_map, A2, A3, A4, A5, B2, B3, B4, B5, temperature=0.001 * 5,
                    detach_flag=False, WTA_scale_weight=1, feature_noise=0, taps=None):
    """WarpNet.forward, models/NonlocalNet.py:427-502.  `taps` (dict) receives intermediates."""
    A_features = warp_features(sd, A2, A3, A4, A5)
    B_features = warp_features(sd, B2, B3, B4, B5)
    theta = corr_project(sd, "theta", A_features)
    phi = corr_project(sd, "phi", B_features)
    y, sim, f = correlate(theta, phi, B_lab_map, temperature, WTA_scale_weight)
    if taps is not None:
        taps.update(A_features=A_features, B_features=B_features, theta=theta, phi=phi,
                    y_small=y, sim_small=sim, argmax=f.argmax(-1),
                    top2=torch.topk(f, 2, dim=-1)[0])
    y = F.interpolate(y, scale_factor=4, mode="nearest")
    sim = F.interpolate(sim, scale_factor=4, mode="nearest")
    return y, sim


# ---------------------------------------------------------------------------------- ColorVidNet
def colorvidnet_forward(sd, x, taps=None):
    """ColorVidNet.forward, models/ColorVidNet.py:96-144."""
    def conv(name, t, dil=1):
        return F.conv2d(t, sd[name + ".weight"], sd[name + ".bias"], padding=dil, dilation=dil)

    def ss(name, t):  # depthwise 1x1 stride-2, no bias (ColorVidNet.py:12,16,21)
        return F.conv2d(t, sd[name + ".weight"], None, stride=2, groups=t.shape[1])

    r = F.relu
    c1_1 = r(conv("conv1_1.2", r(conv("conv1_1.0", x))))
    c1_2 = r(conv("conv1_2", c1_1))
    n1 = _inorm(c1_2)
    c2_1 = r(conv("conv2_1", ss("conv1_2norm_ss", n1)))
    c2_2 = r(conv("conv2_2", c2_1))
    n2 = _inorm(c2_2)
    c3_1 = r(conv("conv3_1", ss("conv2_2norm_ss", n2)))
    c3_2 = r(conv("conv3_2", c3_1))
    c3_3 = r(conv("conv3_3", c3_2))
    n3 = _inorm(c3_3)
    c4 = r(conv("conv4_1", ss("conv3_3norm_ss", n3)))
    c4 = r(conv("conv4_2", c4))
    c4 = r(conv("conv4_3", c4))
    c5 = _inorm(c4)
    for nm in ("conv5_1", "conv5_2", "conv5_3"):
        c5 = r(conv(nm, c5, 2))
    c6 = _inorm(c5)
    for nm in ("conv6_1", "conv6_2", "conv6_3"):
        c6 = r(conv(nm, c6, 2))
    c7 = _inorm(c6)
    for nm in ("conv7_1", "conv7_2", "conv7_3"):
        c7 = r(conv(nm, c7))
    n7 = _inorm(c7)
    c8 = r(conv("conv8_1.1", _up2(n7)) + conv("conv3_3_short", n3))
    c8 = r(conv("conv8_2", c8))
    c8 = r(conv("conv8_3", c8))
    n8 = _inorm(c8)
    c9 = r(conv("conv9_1.1", _up2(n8)) + conv("conv2_2_short", n2))
    c9 = r(conv("conv9_2", c9))
    n9 = _inorm(c9)
    c10 = r(conv("conv10_1.1", _up2(n9)) + conv("conv1_2_short", n1))
    c10 = F.leaky_relu(conv("conv10_2", c10), 0.2)
    ab = F.conv2d(c10, sd["conv10_ab.weight"], sd["conv10_ab.bias"])
    if taps is not None:
        taps.update(c1_2=c1_2, c2_2=c2_2, c3_3=c3_3, c7_3=c7, c8_3=c8, c9_2=c9, c10_2=c10)
    return torch.tanh(ab) * 128


# -------------------------------------------------------------------------------- orchestration
VGG_OUT = ["r12", "r22", "r32", "r42", "r52"]


def warp_color(IA_l, IB_lab, features_B, sd_vgg, sd_warp, temperature=0.01, taps=None):
    """models/FrameColor.py:5-38 (feature_noise is unused there; colornet arg is unused)."""
    A_rgb = gray2rgb_batch(IA_l)
    fA = vgg19_forward(sd_vgg, A_rgb, VGG_OUT, preprocess=True)
    nA = [feature_normalize(t) for t in fA[1:]]
    nB = [feature_normalize(t) for t in features_B[1:]]
    y, sim = warpnet_forward(sd_warp, IB_lab, *nA, *nB, temperature=temperature, taps=taps)
    return y, sim, fA


def frame_colorization(IA_lab, IB_lab, IA_last_lab, features_B, sd_vgg, sd_warp, sd_color,
                       temperature=0.01, taps=None):
    """models/FrameColor.py:41-67 with luminance_noise = feature_noise = 0 (test.py:85-95)."""
    IA_l = IA_lab[:, 0:1, :, :]
    nonlocal_BA_lab, similarity_map, features_A = warp_color(
        IA_l, IB_lab, features_B, sd_vgg, sd_warp, temperature=temperature, taps=taps)
    color_input = torch.cat((IA_l, nonlocal_BA_lab[:, 1:3], similarity_map, IA_last_lab), dim=1)
    ab = colorvidnet_forward(sd_color, color_input, taps=taps)
    return ab, nonlocal_BA_lab, features_A


def exemplar_features(IB_lab, sd_vgg):
    """Per-clip exemplar preparation, test.py:61-66."""
    rgb = tensor_lab2rgb(torch.cat((uncenter_l(IB_lab[:, 0:1]), IB_lab[:, 1:3]), dim=1))
    return vgg19_forward(sd_vgg, rgb, VGG_OUT, preprocess=True)


def colorize_clip(frames_lab, IB_lab, sd_vgg, sd_warp, sd_color, temperature=1e-10,
                  frame_propagate=False):
    """The recurrence of test.py:68-96 over a list of Lab frames; returns the ab predictions."""
    feats_B = exemplar_features(IB_lab, sd_vgg)
    last = None
    outs = []
    for IA_lab in frames_lab:
        if last is None:
            last = IB_lab if frame_propagate else torch.zeros_like(IA_lab)
        ab, _, _ = frame_colorization(IA_lab, IB_lab, last, feats_B, sd_vgg, sd_warp, sd_color,
                                      temperature=temperature)
        last = torch.cat((IA_lab[:, 0:1], ab), dim=1)
        outs.append(ab)
    return outs


def to_dtype(sd, dtype):
    return {k: v.to(dtype) for k, v in sd.items()}
