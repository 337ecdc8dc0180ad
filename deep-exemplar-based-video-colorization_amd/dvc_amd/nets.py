"""Drop-in nn.Module classes for the hot path, backed by the HIP kernels.

Same constructor arguments, forward signatures, return conventions and state_dict keys as
  VGG19_pytorch  /root/reference/models/NonlocalNet.py:192-256
  WarpNet        /root/reference/models/NonlocalNet.py:355-502
  ColorVidNet    /root/reference/models/ColorVidNet.py:6-144
so `load_state_dict(torch.load(...))`, `.parameters()`, `.eval()`, `.cuda()` behave as in
/root/reference/test.py:147-166.  The torch.nn submodules created here are *parameter containers
only* (they give the parameters their reference names); no torch.nn forward is ever called — every
forward pass is a sequence of libdvc_hip.so kernel launches on the current stream.

Inference only: outputs never carry autograd history (the reference's training path,
train.py:402-427, is out of scope — SURVEY.md §2 rows 12-13).
"""
import torch
import torch.nn as nn

from . import arch, ops

_VGG_MEAN_BGR = (0.40760392, 0.45795686, 0.48501961)  # utils/util.py:351


_pack_epoch = 0


def pack_epoch():
    """Number of weight (re)packs so far in this process: a captured launch sequence (dvc_amd/graph.py) bakes the packed
    tensors' addresses in and is re-captured when this has moved (load_state_dict, .cuda(), another conv algorithm)."""
    return _pack_epoch


class _PackCache:
    """Repacked weights ([Cin][k*k][Cout]) keyed by parameter identity + version, so an in-place
    `load_state_dict` or a `.cuda()` move is picked up on the next forward."""

    def __init__(self):
        self._d = {}

    def get(self, key, param, fn):
        """`param`: a parameter, or a tuple of parameters (fn then receives the tuple) — e.g. the concatenated filters of two
        convolutions that run as one launch."""
        params = param if isinstance(param, tuple) else (param,)
        tag = tuple((p.data_ptr(), _version_of(p), str(p.device)) for p in params)
        param = params[0]
        hit = self._d.get(key)
        if hit is None or hit[0] != tag:
            # Cold path (first use, or after load_state_dict / .cuda()).  The packed tensor is produced
            # on whatever stream is current and then read by kernels on ANY stream (the clip driver runs
            # front ends on side streams), so the miss is made a device-wide ordering point: everything
            # that may still read the entry being replaced has finished before it is dropped, and the new
            # entry is complete before any stream can see it.  `prepare()` takes all misses up front.
            if param.is_cuda:
                torch.cuda.synchronize(param.device)
            global _pack_epoch
            _pack_epoch += 1
            with torch.no_grad():
                hit = (tag, fn(params if len(params) > 1 else param))
            if param.is_cuda:
                torch.cuda.synchronize(param.device)
            self._d[key] = hit
        return hit[1]


def invalidate_weight_caches(*modules):
    """Forget every packed filter (and the exemplar memo) of the given modules: the next forward repacks from the current
    parameter values.  Needed only after writing parameters THROUGH `.data` (`p.data.copy_(...)`), which moves no version
    counter; `load_state_dict`, `.cuda()` and in-place operations on the Parameter itself are noticed without it.  Captured
    launch sequences re-capture (pack_epoch moves); a ClipColorizer's cached exemplar side does not know: call its
    `set_exemplar` again when VGG19 / WarpNet weights were rewritten this way."""
    global _pack_epoch
    for m in modules:
        cache = getattr(m, "_cache", None)
        if isinstance(cache, _PackCache):
            if any(p.is_cuda for p in m.parameters()):
                torch.cuda.synchronize()
            cache._d.clear()
        if getattr(m, "_exemplar_memo", None) is not None:
            object.__setattr__(m, "_exemplar_memo", None)
    _pack_epoch += 1


def _version_of(t):
    """`t._version`, or None for a tensor without a version counter (created under torch.inference_mode())."""
    try:
        return t._version
    except RuntimeError:
        return None


def _same_tensors(a, b):
    """Bit-for-bit equality of two nested tuples of tensors (non-tensor leaves compare with ==)."""
    if isinstance(a, (tuple, list)):
        return isinstance(b, (tuple, list)) and len(a) == len(b) and all(_same_tensors(x, y) for x, y in zip(a, b))
    if isinstance(a, torch.Tensor):
        return isinstance(b, torch.Tensor) and a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(a, b))
    return a == b


def _packs(cache, key, weight):
    """kind -> packed weight of one 3x3 layer: "direct" = [Cin][9][Cout], "winograd" = U = G g G^T (ops.conv3x3)."""
    def get(kind):
        if kind == "winograd":
            return cache.get(key + ":wino", weight, ops.pack_winograd_weight)
        if kind == "ws":
            return cache.get(key + ":ws", weight, ops.pack_ws_weight)
        return cache.get(key, weight, ops.pack_conv_weight)
    return get


def _prepack(cache, key, conv):
    """Both packed forms a 3x3 layer may be asked for under the current algorithm choice (ops.set_conv_algo)."""
    get = _packs(cache, key, conv.weight)
    get("direct")
    co, ci, kh, kw = conv.weight.shape
    if ops.conv_algo() != "direct" and conv.stride[0] == 1 and ops.winograd_eligible(ci, co, kh):
        get("winograd")
    if ops.ws_conv_enabled() and conv.stride[0] == 1 and kh == 3 and ops.ws_eligible(ci, co):
        get("ws")


def _norm_in_place(t, **kw):
    """InstanceNorm (+ what follows) of a convolution output, in place — or, when the convolution left its split-K partial
    sums for this launch to add up (ops.ConvPartials), into a new tensor."""
    return ops.instnorm_apply(t, out=None if isinstance(t, ops.ConvPartials) else t, **kw)


def _check_input(x, name):
    if not x.is_cuda:
        raise RuntimeError(f"{name}: input is on {x.device}; the MI355X HIP path has no CPU fallback "
                           "(move the module and its inputs to the GPU with .cuda())")
    if x.requires_grad and torch.is_grad_enabled():
        # the reference would record an autograd graph here (train.py:402-427); this forward is
        # inference-only, so say so instead of silently returning a tensor without history
        raise NotImplementedError(
            f"{name}: an input requires grad and autograd is enabled, but the HIP forward is inference-only "
            "(no backward kernels). Call it under torch.no_grad() as test.py:83 does, or detach the input; "
            "gradients exist only for the fused correlation, see dvc_amd.corr_autograd.")


# ================================================================================================ VGG19
class VGG19_pytorch(nn.Module):
    """NOTE: input tensor should range in [0,1] (RGB); see NonlocalNet.py:193-195."""

    def __init__(self, pool="max"):
        super().__init__()
        for name, cin, cout in arch.VGG_CONVS:
            setattr(self, name, nn.Conv2d(cin, cout, kernel_size=3, padding=1))
        if pool not in ("max", "avg"):
            raise ValueError("pool must be 'max' or 'avg'")
        self._pool = pool
        self._cache = _PackCache()

    def _packed(self, name, swap_bgr=False):
        conv = getattr(self, name)
        if swap_bgr:  # fold RGB->BGR of vgg_preprocess into conv1_1's input-channel order
            return self._cache.get(name + ":bgr", conv.weight, lambda w: ops.pack_conv_weight(w.flip(1)))
        return self._cache.get(name, conv.weight, ops.pack_conv_weight)

    def _pre_affine(self, N=1):
        """vgg_preprocess (utils/util.py:347-352) as a per-(image, channel) affine on the stored R,G,B channels:
        BGR channel c' = 2-c gets (x - mean[c'])*255 = x*255 - 255*mean[c'].  [N*3] each, built once per batch size."""
        w = self.conv1_1.weight
        sc = self._cache.get(f"pre:scale:{N}", w, lambda w: torch.full((3 * N,), 255.0, device=w.device))
        sh = self._cache.get(f"pre:shift:{N}", w, lambda w: torch.tensor(
            [-255.0 * _VGG_MEAN_BGR[2], -255.0 * _VGG_MEAN_BGR[1], -255.0 * _VGG_MEAN_BGR[0]] * N, device=w.device))
        return sc, sh

    def prepare(self):
        """Pack every weight now, on the current stream (see _PackCache.get)."""
        for name, _, _ in arch.VGG_CONVS:
            _prepack(self._cache, name, getattr(self, name))
        self._packed("conv1_1", swap_bgr=True)
        self._pre_affine()

    def forward_gray(self, IA_l, out_keys):
        """forward(gray2rgb_batch(IA_l), out_keys, preprocess=True) — the call FrameColor.py:8-10 makes — with the grey-to-RGB
        replication folded into conv1_1's load (DVC_CONV_GRAY_INPUT): IA_l is the centred luminance [N,1,H,W] (a channel slice
        of the Lab frame is fine).  Bit-identical to the two-step form, one launch and a 1 MB tensor less per frame."""
        return self.forward(IA_l, out_keys, preprocess=True, _gray=True)

    def forward(self, x, out_keys, preprocess=True, _gray=False):
        _check_input(x, "VGG19_pytorch")
        x = x.detach().float() if _gray else x.detach().contiguous().float()
        N = x.shape[0]
        for k in out_keys:
            if k not in arch.VGG_KEYS:
                raise KeyError(k)
        last = max(arch.VGG_KEYS.index(k) for k in out_keys) if out_keys else -1
        out = {}
        cur = x
        conv_names = {("r%s" % n[4:].replace("_", "")): n for n, _, _ in arch.VGG_CONVS}
        pooled = None
        for i, key in enumerate(arch.VGG_KEYS):
            if i > last:
                break  # the reference always runs through p5; the requested outputs are identical
            if key[0] == "p":
                if pooled is not None:          # came out of the convolution in front of it (ops.conv2d_winograd_pool)
                    cur, pooled = pooled, None
                else:
                    cur = ops.maxpool2x2(cur) if self._pool == "max" else ops.avgpool2x2(cur)
            elif (self._pool == "max" and ops.pool_fusion() and i + 1 <= last and arch.VGG_KEYS[i + 1][0] == "p"
                  and min(cur.shape[2:]) >= 2
                  and ops.winograd_selected(N, cur.shape[1], cur.shape[2], cur.shape[3], getattr(self, conv_names[key]).weight.shape[0],
                                            layer="vgg." + conv_names[key])):
                # relu1_2 / relu2_2 / relu3_4 / relu4_4 -> pool: the pooled tensor comes out of the convolution's own launch; the
                # full-resolution one only when somebody asked for it
                conv = getattr(self, conv_names[key])
                if ops.layer_record is not None:
                    ops.layer_record.append(dict(layer="vgg." + conv_names[key], Cin=cur.shape[1], Cout=conv.weight.shape[0],
                                                 H=cur.shape[2], W=cur.shape[3], dil=1, in_up=1, in_sub=1, eligible=True))
                cur, pooled = ops.conv2d_winograd_pool(cur, _packs(self._cache, conv_names[key], conv.weight)("winograd"),
                                                       conv.bias.detach(), act=ops.ACT_RELU, want_full=key in out_keys)
            else:
                name = conv_names[key]
                conv = getattr(self, name)
                bias = conv.bias.detach()
                if name == "conv1_1" and preprocess:
                    # vgg_preprocess folded into the load (BGR weight flip + per-channel affine)
                    sc, sh = self._pre_affine(N)
                    cur = ops.conv2d(cur, self._packed(name, swap_bgr=True), bias, act=ops.ACT_RELU,
                                     in_scale=sc, in_shift=sh, gray_input=_gray)
                else:
                    cur = ops.conv3x3(cur, conv.weight, _packs(self._cache, name, conv.weight), bias, act=ops.ACT_RELU,
                                      layer="vgg." + name)
            out[key] = cur
        return [out[key] for key in out_keys]


# ============================================================================================== WarpNet
class _ResidualBlockParams(nn.Module):
    """Parameter container with the reference's names (conv1, conv2, prelu), NonlocalNet.py:330-339."""

    def __init__(self, ch):
        super().__init__()
        self.conv1 = nn.Conv2d(ch, ch, kernel_size=3, padding=0, stride=1)
        self.conv2 = nn.Conv2d(ch, ch, kernel_size=3, padding=0, stride=1)
        self.prelu = nn.PReLU()


def _head_container(spec):
    n = max(max(ci, pi) for (ci, _, _, _, pi) in spec["convs"]) + 1
    if spec["up_out"]:
        n += 1
    mods = [nn.Identity() for _ in range(n)]
    for (ci, cin, cout, stride, pi) in spec["convs"]:
        mods[ci] = nn.Conv2d(cin, cout, kernel_size=3, padding=0, stride=stride)
        mods[pi] = nn.PReLU()
    return nn.Sequential(*mods)


class WarpNet(nn.Module):
    """ input is Al, Bl, channel = 1, range~[0,255] (docstring of NonlocalNet.py:356) """

    def __init__(self, batch_size):
        super().__init__()
        self.feature_channel = arch.WARP_FEATURE_CH
        self.in_channels = self.feature_channel * 4
        self.inter_channels = arch.WARP_TRUNK_CH
        for name in arch.WARP_HEAD_ORDER:
            setattr(self, name, _head_container(arch.WARP_HEADS[name]))
        self.layer = nn.Sequential(*[_ResidualBlockParams(arch.WARP_TRUNK_CH)
                                     for _ in range(arch.WARP_NUM_RESBLOCKS)])
        self.theta = nn.Conv2d(self.in_channels, self.inter_channels, kernel_size=1, stride=1, padding=0)
        self.phi = nn.Conv2d(self.in_channels, self.inter_channels, kernel_size=1, stride=1, padding=0)
        self._cache = _PackCache()
        # "fp32": exact-fp32 MFMA affinities.  "bf16": bf16 MFMA candidate filter + exact fp32 re-scoring
        # (BASELINE configs[4]); exact for temperature <= 1e-4, otherwise the fp32 kernel is used.
        self.corr_precision = "fp32"

    # -- helpers
    def _use_bf16(self, temperature, WTA_scale_weight):
        return self.corr_precision == "bf16" and float(temperature) <= 1e-4 and WTA_scale_weight == 1

    def _pk(self, key, conv):
        return self._cache.get(key, conv.weight, ops.pack_conv_weight)

    def _conv3(self, key, conv, x, **kw):
        return ops.conv3x3(x, conv.weight, _packs(self._cache, key, conv.weight), conv.bias.detach(), layer="warp." + key, **kw)

    def prepare(self):
        """Pack every weight now, on the current stream (see _PackCache.get)."""
        for name in arch.WARP_HEAD_ORDER:
            seq = getattr(self, name)
            for (ci, _, _, _, _) in arch.WARP_HEADS[name]["convs"]:
                _prepack(self._cache, f"{name}.{ci}", seq[ci])
        for b in range(arch.WARP_NUM_RESBLOCKS):
            _prepack(self._cache, f"layer.{b}.conv1", self.layer[b].conv1)
            _prepack(self._cache, f"layer.{b}.conv2", self.layer[b].conv2)
        self._pk("theta", self.theta)
        self._pk("phi", self.phi)

    def features(self, r2, r3, r4, r5):
        """Heads + concat + residual trunk for one side (NonlocalNet.py:451-465) -> [N,256,h,w]."""
        feats = [r2, r3, r4, r5]
        N = r2.shape[0]
        dev = r2.device
        # geometry of the four head outputs
        shapes = []
        for name, x in zip(arch.WARP_HEAD_ORDER, feats):
            spec = arch.WARP_HEADS[name]
            H, W = x.shape[2], x.shape[3]
            if spec["up_mid"]:
                H, W = 2 * H, 2 * W
            s = spec["convs"][1][3]
            H, W = (H - 1) // s + 1, (W - 1) // s + 1
            if spec["up_out"]:
                H, W = 2 * H, 2 * W
            shapes.append((H, W))
        h, w = shapes[0]
        rpad5 = 0
        if shapes[3] != shapes[0]:  # NonlocalNet.py:461-463 pads one replicated row top and bottom
            rpad5 = 1
            shapes[3] = (shapes[3][0] + 2, shapes[3][1])
        for nm, sh in zip(arch.WARP_HEAD_ORDER, shapes):
            if sh != (h, w):
                raise RuntimeError(f"Sizes of tensors must match except in dimension 1: {nm} gives {sh}, "
                                   f"layer2_1 gives {(h, w)}")
        trunk = torch.empty((N, arch.WARP_TRUNK_CH, h, w), device=dev, dtype=torch.float32)
        bs = arch.WARP_TRUNK_CH * h * w
        # The four heads are mutually independent (NonlocalNet.py:451-458): they advance stage by stage — first convolutions,
        # their norms, second convolutions, final norms into the trunk's channel slices — and each stage is ONE launch over the
        # heads that take the same kind of kernel (ops.conv3x3_group / ops.instnorm_apply_group: bit-identical to per-layer
        # launches, which DVC_GROUP_HEADS=0 brings back).  The stride-2 head (layer2_1, run-time-geometry direct kernel with
        # the norm applied on load) keeps its own launches.
        heads = []
        for i, (name, x) in enumerate(zip(arch.WARP_HEAD_ORDER, feats)):
            spec = arch.WARP_HEADS[name]
            seq = getattr(self, name)
            (ia, _, _, _, pa), (ib, _, _, sb, pb) = spec["convs"]
            heads.append(dict(i=i, name=name, spec=spec, seq=seq, ia=ia, pa=pa, ib=ib, sb=sb, pb=pb, ca=seq[ia], cb=seq[ib], x=x))

        def conv_item(key, conv, x, **kw):
            return dict(x=x, weight=conv.weight, packs=_packs(self._cache, key, conv.weight), bias=conv.bias.detach(),
                        layer="warp." + key, pad_mode=ops.PAD_REFLECT, **kw)

        # stage 1: first convolutions
        t1 = ops.conv3x3_group([conv_item(f"{hd['name']}.{hd['ia']}", hd["ca"], hd["x"], defer_reduce=hd["sb"] == 1) for hd in heads])
        # stage 2: InstanceNorm + PReLU, materialised (one launch for the four heads, in place where the convolution wrote a
        # tensor) so that the next convolution has no fused input transform.  r06: the stride-2 head too — up to r05 its
        # statistics were a launch of their own and the stride-2 convolution applied them on load; as an item of the grouped
        # launch the norm costs nothing extra, and the convolution (run-time-geometry kernel, register staging either way)
        # reads the normalised tensor
        normed = ops.instnorm_apply_group([dict(x=t1[k], out=None if isinstance(t1[k], ops.ConvPartials) else t1[k],
                                                slope_t=hd["seq"][hd["pa"]].weight.detach()) for k, hd in enumerate(heads)])
        plain = [k for k, hd in enumerate(heads) if hd["sb"] == 1]
        t2 = [None] * len(heads)
        for k, hd in enumerate(heads):      # (before the grouped launch: a split-K direct convolution uses the same workspace)
            if hd["sb"] != 1:
                t2[k] = ops.conv2d(normed[k], self._pk(f"{hd['name']}.{hd['ib']}", hd["cb"]), hd["cb"].bias.detach(), stride=hd["sb"],
                                   pad_mode=ops.PAD_REFLECT, in_up=2 if hd["spec"]["up_mid"] else 1)
        # stage 3: second convolutions
        second = ops.conv3x3_group([conv_item(f"{heads[k]['name']}.{heads[k]['ib']}", heads[k]["cb"], normed[k],
                                              in_up=2 if heads[k]["spec"]["up_mid"] else 1, defer_reduce=True)
                                    for k in plain])
        for j, k in enumerate(plain):
            t2[k] = second[j]
        # stage 4: final norms (+ PReLU, x2 upsample, replicated rows) into the trunk's channel slices
        ops.instnorm_apply_group([dict(x=t2[k], slope_t=hd["seq"][hd["pb"]].weight.detach(), up=2 if hd["spec"]["up_out"] else 1,
                                       rpad=rpad5 if hd["name"] == "layer5_1" else 0,
                                       out=trunk[:, hd["i"] * arch.WARP_FEATURE_CH:(hd["i"] + 1) * arch.WARP_FEATURE_CH],
                                       out_batch_stride=bs) for k, hd in enumerate(heads)])
        x = trunk
        for b in range(arch.WARP_NUM_RESBLOCKS):
            blk = self.layer[b]
            a = blk.prelu.weight.detach()
            t = self._conv3(f"layer.{b}.conv1", blk.conv1, x, pad_mode=ops.PAD_REFLECT, defer_reduce=True)
            t = _norm_in_place(t, slope_t=a)
            t = self._conv3(f"layer.{b}.conv2", blk.conv2, t, pad_mode=ops.PAD_REFLECT, defer_reduce=True)
            x = _norm_in_place(t, residual=x, slope_t=a)
        return x

    def project(self, which, feats, bf16=False):
        """theta / phi: 1x1 conv, centre over positions, L2-normalise over channels -> [N,256,P]
        (bf16=True: the ([N,P,256] fp32, [N,P,256] bf16) pair the bf16 correlation consumes)."""
        conv = getattr(self, which)
        t = ops.conv2d(feats, self._pk(which, conv), conv.bias.detach(), ksize=1, pad=0)
        return ops.corr_prepare_bf16(t) if bf16 else ops.corr_prepare(t)

    # ---- transparent exemplar memo (r05).  The reference's loop (test.py:68-96) hands the SAME exemplar tensors to every
    # frame_colorization call and NonlocalNet.py:452-465,473-476,491-493 recompute the exemplar side each time.  A caller that is
    # not rewritten around ClipColorizer gets the cached form anyway: the exemplar side is memoised on the identity and the
    # version counters of the tensors it was computed from (the memo holds references to them, so their addresses cannot be
    # recycled for other data while it is alive), this module's parameters (data_ptr, _version) and everything that selects
    # kernels.  An in-place write to a key tensor bumps its `_version` and misses.  What no version counter sees:
    #   * tensors created under torch.inference_mode() have no version counter at all — the memo is bypassed for them (the
    #     exemplar side is recomputed per call, as the reference does);
    #   * writes through `.data` (`t.data.copy_(...)`, `p.data.mul_(...)`) are invisible here as they are to autograd: NOT
    #     supported with the memo on.  DVC_EXEMPLAR_MEMO=verify / ops.set_exemplar_memo("verify")
    #     recomputes on every hit, compares bit for bit, warns on a mismatch and returns the fresh value.
    # DVC_EXEMPLAR_MEMO=0 / ops.set_exemplar_memo(False) turns the memo off.
    def _memo_exemplar_side(self, key_tensors, regime, compute):
        mode = ops.exemplar_memo_mode()
        if mode == "off":
            return compute()
        versions = [_version_of(t) for t in key_tensors]
        pfp = tuple((p.data_ptr(), _version_of(p)) for p in self.parameters())
        if any(v is None for v in versions) or any(v is None for _, v in pfp):
            return compute()        # inference tensors: nothing to key a change on
        fp = (pfp, regime, ops.conv_algo(), ops.direct_layers(), ops.fuse_reduce(), ops.autotune_enabled(),
              ops.batch_plan_enabled(), ops.group_heads(), ops.ws_conv_enabled())
        memo = getattr(self, "_exemplar_memo", None)
        if (memo is not None and memo[1] == fp and len(memo[0]) == len(key_tensors)
                and all(a is b and va == vb for (a, va), b, vb in zip(memo[0], key_tensors, versions))):
            if mode != "verify":
                return memo[2]
            fresh = compute()
            if _same_tensors(fresh, memo[2]):
                return memo[2]
            import warnings
            warnings.warn("DVC_EXEMPLAR_MEMO=verify: the memoised exemplar side differs from a recomputation although no key "
                          "tensor or WarpNet parameter changed identity or version — something wrote to them through `.data` "
                          "(unsupported with the memo on); using the recomputed value", RuntimeWarning, stacklevel=3)
            value = fresh
        else:
            value = compute()
        # (object.__setattr__: the memo holds tensors, nn.Module.__setattr__ would try to register them)
        object.__setattr__(self, "_exemplar_memo", ([(t, v) for t, v in zip(key_tensors, versions)], fp, value))
        return value

    def __getstate__(self):
        # (pickling / copy.deepcopy of the module: the memo is a cache of tensors, not state)
        state = dict(self.__dict__)
        state.pop("_exemplar_memo", None)
        return state

    def exemplar_side(self, B_lab_map, B2, B3, B4, B5, bf16=None):
        """Everything that depends only on the exemplar (recomputed per frame by the reference,
        NonlocalNet.py:452-465,473-476,491-493; cacheable per clip)."""
        if bf16 is None:
            bf16 = self.corr_precision == "bf16"
        phi = self.project("phi", self.features(B2, B3, B4, B5), bf16=bf16)
        blab = ops.avgpool4x4(B_lab_map)
        return phi, blab

    def forward(self, B_lab_map, A_relu2_1, A_relu3_1, A_relu4_1, A_relu5_1, B_relu2_1, B_relu3_1,
                B_relu4_1, B_relu5_1, temperature=0.001 * 5, detach_flag=False, WTA_scale_weight=1,
                feature_noise=0, exemplar_cache=None, return_taps=False, defer_merge=False):
        """models/NonlocalNet.py:427-502 -> (y, similarity_map).  Not upstream: `exemplar_cache` (the exemplar side computed once
        per clip), `return_taps`, and `defer_merge` — the fp32 correlation then leaves the merge of its partial softmax states
        to the consumer and the call returns (ops.CorrPartials, None) for ops.pack_color_input (dvc_amd/frame.py)."""
        ins = [B_lab_map, A_relu2_1, A_relu3_1, A_relu4_1, A_relu5_1, B_relu2_1, B_relu3_1, B_relu4_1,
               B_relu5_1]
        for t in ins:
            _check_input(t, "WarpNet")
        memo_key = (B_lab_map, B_relu2_1, B_relu3_1, B_relu4_1, B_relu5_1)      # (the caller's own tensor objects)
        ins = [t.detach().contiguous().float() for t in ins]
        B_lab_map, A2, A3, A4, A5, B2, B3, B4, B5 = ins
        image_height, image_width = B_lab_map.shape[2], B_lab_map.shape[3]
        fh, fw = int(image_height / 4), int(image_width / 4)
        A_features = self.features(A2, A3, A4, A5)
        if (A_features.shape[2], A_features.shape[3]) != (fh, fw):
            raise RuntimeError(f"shape '[{B_lab_map.shape[0]}, 1, {fh}, {fw}]' is invalid for feature map "
                               f"of size {tuple(A_features.shape[2:])}")
        bf16 = self._use_bf16(temperature, WTA_scale_weight)
        if exemplar_cache is not None:
            phi, blab = exemplar_cache
            if isinstance(phi, tuple) != bf16:
                raise RuntimeError("exemplar cache was built for a different corr_precision / temperature regime")
        else:
            phi, blab = self._memo_exemplar_side(memo_key, ("forward", bf16),
                                                 lambda: self.exemplar_side(B_lab_map, B2, B3, B4, B5, bf16=bf16))
        theta = self.project("theta", A_features, bf16=bf16)
        if bf16:
            res = ops.corr_fwd_bf16(theta, phi, blab.view(blab.shape[0], 3, -1), float(temperature), fh, fw,
                                    want_small=return_taps, want_argmax=return_taps)
        else:
            res = ops.corr_fwd(theta, phi, blab.view(blab.shape[0], 3, -1), float(temperature), fh, fw,
                               wta_scale=float(WTA_scale_weight), want_small=return_taps,
                               want_argmax=return_taps, defer_merge=defer_merge and not return_taps)
            if isinstance(res, ops.CorrPartials):
                return res, None
        if return_taps:
            return res["y_up"], res["sim_up"], dict(theta=theta, phi=phi, y_small=res["y_small"],
                                                    sim_small=res["sim_small"], argmax=res["argmax"],
                                                    A_features=A_features)
        return res["y_up"], res["sim_up"]


# ========================================================================================== ColorVidNet
class ColorVidNet(nn.Module):
    def __init__(self, ic):
        super().__init__()
        shapes = arch.colorvidnet_param_shapes(ic)
        made = {}
        for key, kind in arch.CVN_STATE_ORDER:
            w = shapes[key + ".weight"]
            if kind == "ss":
                mod = nn.Conv2d(w[0], w[0], 1, 2, bias=False, groups=w[0])
            else:
                k = w[2]
                mod = nn.Conv2d(w[1], w[0], k, 1, 0)
            made[key] = mod
        # group "a.b" keys into Sequential containers so that state_dict keys match the reference
        groups = {}
        for key, mod in made.items():
            if "." in key:
                top, idx = key.split(".")
                groups.setdefault(top, {})[int(idx)] = mod
            else:
                setattr(self, key, mod)
        for top, items in groups.items():
            seq = [nn.Identity() for _ in range(max(items) + 1)]
            for i, m in items.items():
                seq[i] = m
            setattr(self, top, nn.Sequential(*seq))
        # keep the reference's registration order (ColorVidNet.py:9-47) for state_dict()
        order = []
        for key, _ in arch.CVN_STATE_ORDER:
            top = key.split(".")[0]
            if top not in order:
                order.append(top)
        self._modules = type(self._modules)((k, self._modules[k]) for k in order)
        self._ic = ic
        self._cache = _PackCache()
        # the reference constructor prints these two lines (ColorVidNet.py:80,85)
        print("replace all deconv with [nearest + conv]")
        print("replace all batchnorm with instancenorm")

    def _mod(self, key):
        m = self
        for part in key.split("."):
            m = m[int(part)] if part.isdigit() else getattr(m, part)
        return m

    def _ss_weight(self, key):
        return self._cache.get(key, self._mod(key).weight, lambda w: w.detach().reshape(-1).contiguous())

    def _out_weight(self):
        out = self._mod(arch.CVN_OUT["key"])
        return self._cache.get("conv10_ab", out.weight, lambda w: w.detach().reshape(w.shape[0], -1).contiguous())

    # ---- decoder blocks: `conv8_1(up(norm(c7_3))) + conv3_3_short(norm(c3_3))` (ColorVidNet.py:124-127; likewise conv9_1,
    # conv10_1) as ONE launch over the channels of both inputs (ops.conv2d_winograd_dual) when both are Winograd layers
    def _dual_pairs(self):
        """consumer key -> the skip-convolution entry whose output it adds (a linear convolution read by nothing else)."""
        by_dst = {c["dst"]: c for c in arch.CVN_CONVS}
        uses = {}
        for c in arch.CVN_CONVS:
            uses[c["src"]] = uses.get(c["src"], 0) + 1
            if c["add"] is not None:
                uses[c["add"]] = uses.get(c["add"], 0) + 1
        pairs = {}
        for c in arch.CVN_CONVS:
            e = by_dst.get(c["add"]) if c["add"] is not None else None
            if e is not None and e["act"] == "none" and e["add"] is None and uses.get(e["dst"], 0) == 1 and e["dil"] == c["dil"] \
                    and e["pre"] in (None, "norm") and c["pre"] in ("up", "norm", None):
                pairs[c["key"]] = e
        return pairs

    def _dual_pack(self, cA, cB):
        """(concatenated Winograd filters, summed bias) of a fused pair; cached per parameter versions like every pack."""
        mA, mB = self._mod(cA["key"]), self._mod(cB["key"])
        u = self._cache.get(cA["key"] + ":dual", (mA.weight, mB.weight), lambda ws: torch.cat(
            (ops.pack_winograd_weight(ws[0]), ops.pack_winograd_weight(ws[1])), dim=1).contiguous())
        b = self._cache.get(cA["key"] + ":dualbias", (mA.bias, mB.bias), lambda bs: (bs[0].detach() + bs[1].detach()).contiguous())
        return u, b

    def _dual_ok(self, cA, cB, shapeA, shapeB):
        """Both convolutions of the pair go to the Winograd kernel under the current algorithm choice."""
        if not ops.dual_conv_enabled():
            return False
        (N, CA, HA, WA), (_, CB, HB, WB) = shapeA, shapeB
        upA = 2 if cA["pre"] == "up" else 1
        # (dvc_conv2d_winograd_dual forces the 64-channel x 32-tile workgroup shape and stages 8-channel chunks of each input)
        if cA["cout"] % 64 or cB["cout"] != cA["cout"] or CA % 8 or CB % 8:
            return False
        return (ops.winograd_selected(N, CA, HA, WA, cA["cout"], dil=cA["dil"], pad=cA["dil"], in_up=upA, layer="cvn." + cA["key"])
                and ops.winograd_selected(N, CB, HB, WB, cB["cout"], dil=cB["dil"], pad=cB["dil"], layer="cvn." + cB["key"]))

    def prepare(self):
        """Pack every weight now, on the current stream (see _PackCache.get)."""
        for c in arch.CVN_CONVS:
            _prepack(self._cache, c["key"], self._mod(c["key"]))
            if c["pre"] == "norm_ss":
                self._ss_weight(c["ss"])
        self._out_weight()
        if ops.conv_algo() != "direct" and ops.dual_conv_enabled():
            by_key = {c["key"]: c for c in arch.CVN_CONVS}
            for key, e in self._dual_pairs().items():
                self._dual_pack(by_key[key], e)

    def forward(self, x):
        """ x: gray image (1 channel), ab(2 channel), ab_err, ba_err"""
        _check_input(x, "ColorVidNet")
        x = x.detach().contiguous().float()
        acts = {"x": x}
        normed = {}
        ss_weight = self._ss_weight

        # activations that are normalised for two consumers (skip convolution: plain; next block: * `_ss`
        # weight, stride 2) get both tensors from one launch
        both = {c["src"]: c["ss"] for c in arch.CVN_CONVS if c["pre"] == "norm_ss"}
        both = {k: v for k, v in both.items() if any(c["src"] == k and c["pre"] in ("norm", "up") for c in arch.CVN_CONVS)}

        def norm_of(src, ss_key=None):
            """InstanceNorm2d(src) [* the depthwise `_ss` weight, stride 2] as a tensor (ColorVidNet.py:85-94,12)."""
            k = (src, ss_key)
            if k not in normed:
                if src in both:
                    normed[(src, None)], normed[(src, both[src])] = ops.instnorm_apply(
                        acts[src], eps=1e-5, second=(ss_weight(both[src]), 2))
                else:
                    normed[k] = ops.instnorm_apply(acts[src], eps=1e-5,
                                                   chan_scale=ss_weight(ss_key) if ss_key else None,
                                                   sub=2 if ss_key else 1)
            return normed[k]

        # activations that are only ever read through an InstanceNorm: normalised right after the convolution that
        # produces them, which may then leave its split-K partial sums for the InstanceNorm launch to add up
        norm_uses = {}
        raw_use = set()
        for c in arch.CVN_CONVS:
            if c["pre"] in ("norm", "norm_ss", "up"):
                norm_uses.setdefault(c["src"], []).append(c["ss"] if c["pre"] == "norm_ss" else None)
            else:
                raw_use.add(c["src"])
            if c["add"] is not None:
                raw_use.add(c["add"])
        raw_use.add(arch.CVN_OUT.get("src", "c10_2"))

        act_map = {"relu": ops.ACT_RELU, "none": ops.ACT_NONE, "leaky": ops.ACT_LEAKY}
        pairs = self._dual_pairs()
        deferred = {e["key"] for e in pairs.values()}
        for c in arch.CVN_CONVS:
            if c["key"] in deferred:
                continue            # the skip convolution of a decoder block: runs with its consumer below (or just before it)
            e = pairs.get(c["key"])
            if e is not None:
                srcA = norm_of(c["src"]) if c["pre"] in ("norm", "up") else acts[c["src"]]
                srcB = norm_of(e["src"]) if e["pre"] == "norm" else acts[e["src"]]
                if self._dual_ok(c, e, srcA.shape, srcB.shape):
                    u, b = self._dual_pack(c, e)
                    if ops.layer_record is not None:
                        for cc_, src_ in ((c, srcA), (e, srcB)):
                            ops.layer_record.append(dict(layer="cvn." + cc_["key"], Cin=src_.shape[1], Cout=cc_["cout"], H=src_.shape[2],
                                                         W=src_.shape[3], dil=cc_["dil"], in_up=2 if cc_["pre"] == "up" else 1, in_sub=1,
                                                         eligible=True, dual=c["key"]))
                    acts[c["dst"]] = ops.conv2d_winograd_dual(srcA, srcB, u, b, dil=c["dil"], in_upA=2 if c["pre"] == "up" else 1,
                                                              act=act_map[c["act"]], act_slope=0.2)
                    if c["dst"] in norm_uses:
                        for ss_key in dict.fromkeys(norm_uses[c["dst"]]):
                            norm_of(c["dst"], ss_key)
                    continue
                # not both Winograd layers (direct algorithm, tiny maps): the skip convolution as its own launch, then the adder
                convE = self._mod(e["key"])
                acts[e["dst"]] = ops.conv3x3(srcB, convE.weight, _packs(self._cache, e["key"], convE.weight), convE.bias.detach(),
                                             dil=e["dil"], act=act_map[e["act"]], act_slope=0.2, layer="cvn." + e["key"])
            conv = self._mod(c["key"])
            kw = dict(dil=c["dil"], act=act_map[c["act"]], act_slope=0.2)
            pre = c["pre"]
            src = acts[c["src"]]
            if pre == "norm":
                src = norm_of(c["src"])
            elif pre == "norm_ss":
                src = norm_of(c["src"], c["ss"])
            elif pre == "up":
                src = norm_of(c["src"])
                kw["in_up"] = 2
            if c["add"] is not None:
                kw["residual"] = acts[c["add"]]
            dst = c["dst"]
            acts[dst] = ops.conv3x3(src, conv.weight, _packs(self._cache, c["key"], conv.weight), conv.bias.detach(),
                                    defer_reduce=dst in norm_uses and dst not in raw_use, layer="cvn." + c["key"], **kw)
            if dst in norm_uses:
                if dst in both:
                    norm_of(dst)
                else:
                    for ss_key in dict.fromkeys(norm_uses[dst]):
                        norm_of(dst, ss_key)
        out = self._mod(arch.CVN_OUT["key"])
        return ops.conv1x1_small(acts["c10_2"], self._out_weight(), out.bias.detach(), act=ops.ACT_TANH128)
