"""Static architecture tables for the three networks on the hot path.

These tables are the single source of truth for (a) the state_dict key contract the drop-in
modules must honour, (b) the flat parameter-buffer layout that the C-ABI
library consumes (include/dvc_hip.h, "parameter packing"), and (c) the synthetic weight
factory in ``synth.py``.

Shapes follow the reference constructors:
  VGG19_pytorch   /root/reference/models/NonlocalNet.py:197-226
  WarpNet         /root/reference/models/NonlocalNet.py:358-425
  ColorVidNet     /root/reference/models/ColorVidNet.py:7-94
"""

# ---------------------------------------------------------------------------------------------
# VGG19: 16 conv3x3 (pad 1) + ReLU, 5 maxpool2x2.  (name, cin, cout)
VGG_CONVS = [
    ("conv1_1", 3, 64), ("conv1_2", 64, 64),
    ("conv2_1", 64, 128), ("conv2_2", 128, 128),
    ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv3_4", 256, 256),
    ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), ("conv4_4", 512, 512),
    ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512), ("conv5_4", 512, 512),
]
# forward order of activation keys (NonlocalNet.py:235-256); "p*" are the pools
VGG_KEYS = [
    "r11", "r12", "p1", "r21", "r22", "p2", "r31", "r32", "r33", "r34", "p3",
    "r41", "r42", "r43", "r44", "p4", "r51", "r52", "r53", "r54", "p5",
]

# ---------------------------------------------------------------------------------------------
# WarpNet feature heads: list of (state_dict conv index, cin, cout, stride, prelu index).
# layer5_1 has an extra Upsample between its two convs so its indices differ
# (NonlocalNet.py:364-410).
WARP_HEADS = {
    "layer2_1": dict(convs=[(1, 128, 128, 1, 3), (5, 128, 64, 2, 7)], up_mid=False, up_out=False),
    "layer3_1": dict(convs=[(1, 256, 128, 1, 3), (5, 128, 64, 1, 7)], up_mid=False, up_out=False),
    "layer4_1": dict(convs=[(1, 512, 256, 1, 3), (5, 256, 64, 1, 7)], up_mid=False, up_out=True),
    "layer5_1": dict(convs=[(1, 512, 256, 1, 3), (6, 256, 64, 1, 8)], up_mid=True, up_out=True),
}
WARP_HEAD_ORDER = ["layer2_1", "layer3_1", "layer4_1", "layer5_1"]
WARP_FEATURE_CH = 64
WARP_TRUNK_CH = 256          # 4 heads x 64 channels, also theta/phi inter_channels
WARP_NUM_RESBLOCKS = 3


def warpnet_param_shapes():
    """Ordered {key: shape} for WarpNet.state_dict() (43 tensors)."""
    out = {}
    for name in WARP_HEAD_ORDER:
        for (ci, cin, cout, _s, pi) in WARP_HEADS[name]["convs"]:
            out[f"{name}.{ci}.weight"] = (cout, cin, 3, 3)
            out[f"{name}.{ci}.bias"] = (cout,)
            out[f"{name}.{pi}.weight"] = (1,)
    for b in range(WARP_NUM_RESBLOCKS):
        for cv in ("conv1", "conv2"):
            out[f"layer.{b}.{cv}.weight"] = (WARP_TRUNK_CH, WARP_TRUNK_CH, 3, 3)
            out[f"layer.{b}.{cv}.bias"] = (WARP_TRUNK_CH,)
        out[f"layer.{b}.prelu.weight"] = (1,)
    for nm in ("theta", "phi"):
        out[f"{nm}.weight"] = (WARP_TRUNK_CH, WARP_TRUNK_CH, 1, 1)
        out[f"{nm}.bias"] = (WARP_TRUNK_CH,)
    return out


def vgg_param_shapes():
    out = {}
    for (name, cin, cout) in VGG_CONVS:
        out[f"{name}.weight"] = (cout, cin, 3, 3)
        out[f"{name}.bias"] = (cout,)
    return out


# ---------------------------------------------------------------------------------------------
# ColorVidNet: conv list in *execution* order.  Each entry:
#   key      state_dict prefix (".weight"/".bias" appended)
#   cin,cout
#   dil      dilation (pad == dil for all 3x3 convs here)
#   src      name of the activation this conv reads
#   pre      how the source is transformed on load:
#              None      as stored
#              "norm"    InstanceNorm of src (eps 1e-5, biased var)
#              "norm_ss" InstanceNorm -> depthwise 1x1 stride-2 scale (`*_ss.weight`)
#              "up"      InstanceNorm -> nearest x2 upsample
#   add      name of an activation added to the conv output before the activation (or None)
#   act      "relu" | "none" | "leaky" | "tanh128"
#   dst      name the result is stored under
# (ColorVidNet.py:96-144)
CVN_CONVS = [
    dict(key="conv1_1.0", cin=None, cout=32, dil=1, src="x", pre=None, add=None, act="relu", dst="c1_1a"),
    dict(key="conv1_1.2", cin=32, cout=64, dil=1, src="c1_1a", pre=None, add=None, act="relu", dst="c1_1"),
    dict(key="conv1_2", cin=64, cout=64, dil=1, src="c1_1", pre=None, add=None, act="relu", dst="c1_2"),
    dict(key="conv2_1", cin=64, cout=128, dil=1, src="c1_2", pre="norm_ss", ss="conv1_2norm_ss", add=None, act="relu", dst="c2_1"),
    dict(key="conv2_2", cin=128, cout=128, dil=1, src="c2_1", pre=None, add=None, act="relu", dst="c2_2"),
    dict(key="conv3_1", cin=128, cout=256, dil=1, src="c2_2", pre="norm_ss", ss="conv2_2norm_ss", add=None, act="relu", dst="c3_1"),
    dict(key="conv3_2", cin=256, cout=256, dil=1, src="c3_1", pre=None, add=None, act="relu", dst="c3_2"),
    dict(key="conv3_3", cin=256, cout=256, dil=1, src="c3_2", pre=None, add=None, act="relu", dst="c3_3"),
    dict(key="conv4_1", cin=256, cout=512, dil=1, src="c3_3", pre="norm_ss", ss="conv3_3norm_ss", add=None, act="relu", dst="c4_1"),
    dict(key="conv4_2", cin=512, cout=512, dil=1, src="c4_1", pre=None, add=None, act="relu", dst="c4_2"),
    dict(key="conv4_3", cin=512, cout=512, dil=1, src="c4_2", pre=None, add=None, act="relu", dst="c4_3"),
    dict(key="conv5_1", cin=512, cout=512, dil=2, src="c4_3", pre="norm", add=None, act="relu", dst="c5_1"),
    dict(key="conv5_2", cin=512, cout=512, dil=2, src="c5_1", pre=None, add=None, act="relu", dst="c5_2"),
    dict(key="conv5_3", cin=512, cout=512, dil=2, src="c5_2", pre=None, add=None, act="relu", dst="c5_3"),
    dict(key="conv6_1", cin=512, cout=512, dil=2, src="c5_3", pre="norm", add=None, act="relu", dst="c6_1"),
    dict(key="conv6_2", cin=512, cout=512, dil=2, src="c6_1", pre=None, add=None, act="relu", dst="c6_2"),
    dict(key="conv6_3", cin=512, cout=512, dil=2, src="c6_2", pre=None, add=None, act="relu", dst="c6_3"),
    dict(key="conv7_1", cin=512, cout=512, dil=1, src="c6_3", pre="norm", add=None, act="relu", dst="c7_1"),
    dict(key="conv7_2", cin=512, cout=512, dil=1, src="c7_1", pre=None, add=None, act="relu", dst="c7_2"),
    dict(key="conv7_3", cin=512, cout=512, dil=1, src="c7_2", pre=None, add=None, act="relu", dst="c7_3"),
    dict(key="conv3_3_short", cin=256, cout=256, dil=1, src="c3_3", pre="norm", add=None, act="none", dst="s3"),
    dict(key="conv8_1.1", cin=512, cout=256, dil=1, src="c7_3", pre="up", add="s3", act="relu", dst="c8_1"),
    dict(key="conv8_2", cin=256, cout=256, dil=1, src="c8_1", pre=None, add=None, act="relu", dst="c8_2"),
    dict(key="conv8_3", cin=256, cout=256, dil=1, src="c8_2", pre=None, add=None, act="relu", dst="c8_3"),
    dict(key="conv2_2_short", cin=128, cout=128, dil=1, src="c2_2", pre="norm", add=None, act="none", dst="s2"),
    dict(key="conv9_1.1", cin=256, cout=128, dil=1, src="c8_3", pre="up", add="s2", act="relu", dst="c9_1"),
    dict(key="conv9_2", cin=128, cout=128, dil=1, src="c9_1", pre=None, add=None, act="relu", dst="c9_2"),
    dict(key="conv1_2_short", cin=64, cout=128, dil=1, src="c1_2", pre="norm", add=None, act="none", dst="s1"),
    dict(key="conv10_1.1", cin=128, cout=128, dil=1, src="c9_2", pre="up", add="s1", act="relu", dst="c10_1"),
    dict(key="conv10_2", cin=128, cout=128, dil=1, src="c10_1", pre=None, add=None, act="leaky", dst="c10_2"),
]
CVN_SS = [("conv1_2norm_ss", 64), ("conv2_2norm_ss", 128), ("conv3_3norm_ss", 256)]
CVN_OUT = dict(key="conv10_ab", cin=128, cout=2)   # 1x1 conv then tanh*128 (ColorVidNet.py:142-144)

# state_dict key order of the reference module (registration order in ColorVidNet.__init__,
# with conv8_1/9_1/10_1 re-registered in place as Sequential(Upsample, Conv2d)).
CVN_STATE_ORDER = [
    ("conv1_1.0", "conv"), ("conv1_1.2", "conv"), ("conv1_2", "conv"), ("conv1_2norm_ss", "ss"),
    ("conv2_1", "conv"), ("conv2_2", "conv"), ("conv2_2norm_ss", "ss"),
    ("conv3_1", "conv"), ("conv3_2", "conv"), ("conv3_3", "conv"), ("conv3_3norm_ss", "ss"),
    ("conv4_1", "conv"), ("conv4_2", "conv"), ("conv4_3", "conv"),
    ("conv5_1", "conv"), ("conv5_2", "conv"), ("conv5_3", "conv"),
    ("conv6_1", "conv"), ("conv6_2", "conv"), ("conv6_3", "conv"),
    ("conv7_1", "conv"), ("conv7_2", "conv"), ("conv7_3", "conv"),
    ("conv8_1.1", "conv"), ("conv3_3_short", "conv"), ("conv8_2", "conv"), ("conv8_3", "conv"),
    ("conv9_1.1", "conv"), ("conv2_2_short", "conv"), ("conv9_2", "conv"),
    ("conv10_1.1", "conv"), ("conv1_2_short", "conv"), ("conv10_2", "conv"), ("conv10_ab", "conv"),
]


def colorvidnet_param_shapes(ic=7):
    by_key = {c["key"]: c for c in CVN_CONVS}
    ss = dict(CVN_SS)
    out = {}
    for key, kind in CVN_STATE_ORDER:
        if kind == "ss":
            out[f"{key}.weight"] = (ss[key], 1, 1, 1)
            continue
        if key == CVN_OUT["key"]:
            out[f"{key}.weight"] = (CVN_OUT["cout"], CVN_OUT["cin"], 1, 1)
            out[f"{key}.bias"] = (CVN_OUT["cout"],)
            continue
        c = by_key[key]
        cin = ic if c["cin"] is None else c["cin"]
        out[f"{key}.weight"] = (c["cout"], cin, 3, 3)
        out[f"{key}.bias"] = (c["cout"],)
    return out


# ---------------------------------------------------------------------------------------------
# Error-aware engine map (ops.direct_layers): the 3x3 layers that stay on the direct implicit-GEMM engine under the default
# algorithm choice because their Winograd rounding moves the frame's ab output the most per microsecond saved
# (tools/engine_sensitivity.py on the MI355X, profiles/r05_engine_sensitivity.txt).  Names = state_dict prefixes with the
# network in front ("vgg.", "warp.", "cvn.").
# What the measurements say (r05, MI355X, 216x384, plain seed-0 weights, T = 1e-10, 10 frames without arg-max flips against the
# fp64 oracle, CPU fp32 = 1.00; profiles/r05_engine_map_eval.txt):
#   every eligible layer on Winograd ("speed")        rms 1.10  mean 1.09  q999 1.16  max 1.92
#   the whole FRONT END direct, ColorVidNet Winograd  rms 1.09  — the front end's engine does not matter: at T = 1e-10 it reaches
#                                                       ColorVidNet only through the similarity map's last bits and the arg-max
#   ColorVidNet direct, front end Winograd            rms 0.81  — the same as everything direct (0.82)
#   conv2_1, conv2_2 direct                           rms 0.99  mean 0.98  q999 1.02  max 1.72
#   conv1_1.2 .. conv2_2 direct                       rms 0.91  mean 0.91  q999 0.94  max 1.26
#   conv1_1.2 .. conv3_3 direct (this map)            rms 0.85  mean 0.85  q999 0.85  max 0.91   (worst frame: max 1.06)
# The excess sits in ColorVidNet's first seven 3x3 layers (an error there is amplified by everything behind it: perturbation
# energy per layer falls 30x from conv1_2 to conv4_3, profiles/r05_engine_sensitivity.txt, while Winograd's own rounding is
# 1.1-2.2x the direct sum's on every layer, profiles/r05_engine_layer_error.txt); they cost 107-135 us of convolution time per
# frame on the direct engine.  Forcing a larger split over input channels on the Winograd kernel (shorter accumulation chains)
# buys less accuracy per microsecond than the direct engine on every one of them (same file).
# r06: re-evaluated with tools/parity_pool_probe.py (8 frames without arg-max flips, profiles/r06_parity_pool_probe_maps.txt) now
# that five of the seven layers run on the weights-in-registers kernel: without conv3_3 (the 256 -> 256 layer at 54x96 right in
# front of an InstanceNorm: as a Winograd launch it also leaves its split-K reduce to that norm's launch) the pooled statistics
# are max 1.01, q99.9 0.90, mean 0.87, rms 0.88 of the CPU fp32 run's — the seven-layer map: 1.03 / 0.90 / 0.86 / 0.86 — for 25 us
# per frame less; without conv3_2 as well q99.9 reaches 1.01 (mean 0.90), so that one stays.
DIRECT_LAYERS = frozenset(("cvn.conv1_1.2", "cvn.conv1_2", "cvn.conv2_1", "cvn.conv2_2", "cvn.conv3_1", "cvn.conv3_2"))
