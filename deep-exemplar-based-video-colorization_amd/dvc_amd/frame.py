"""Frame orchestration: drop-in `warp_color` / `frame_colorization`
(/root/reference/models/FrameColor.py:5-38, 41-67) plus an exemplar-cached clip driver.
"""
import torch

from . import ops
from .nets import _version_of
from .util import feature_normalize, gray2rgb_batch

VGG_OUT = ["r12", "r22", "r32", "r42", "r52"]


def warp_color(IA_l, IB_lab, features_B, vggnet, nonlocal_net, colornet, feature_noise=0, temperature=0.01,
               exemplar_cache=None, defer_merge=False):
    """models/FrameColor.py:5-38.  `colornet` and `feature_noise` are unused there as well.
    defer_merge (not upstream): the correlation's merge is left to ops.pack_color_input — `nonlocal_BA_lab` is then an
    ops.CorrPartials and `similarity_map` None (the fp32 correlation only; the bf16 candidate filter returns tensors)."""
    if ops.gray_fusion() and hasattr(vggnet, "forward_gray") and IA_l.is_cuda:
        # gray2rgb_batch folded into conv1_1's load (bit-identical, one launch less; DVC_GRAY_FUSION=0 restores the two calls)
        A_relu1_1, A_relu2_1, A_relu3_1, A_relu4_1, A_relu5_1 = vggnet.forward_gray(IA_l, VGG_OUT)
    else:
        IA_rgb_from_gray = gray2rgb_batch(IA_l)
        A_relu1_1, A_relu2_1, A_relu3_1, A_relu4_1, A_relu5_1 = vggnet(IA_rgb_from_gray, VGG_OUT, preprocess=True)
    if exemplar_cache is None:
        B_relu1_1, B_relu2_1, B_relu3_1, B_relu4_1, B_relu5_1 = features_B
    # NOTE: output the feature before normalization (FrameColor.py:13-14)
    features_A = [A_relu1_1, A_relu2_1, A_relu3_1, A_relu4_1, A_relu5_1]
    nA = ops.channel_l2norm_multi(features_A[1:])            # feature_normalize x4 (FrameColor.py:16-19), one launch
    if exemplar_cache is None and ops.exemplar_memo_enabled() and hasattr(nonlocal_net, "_memo_exemplar_side"):
        # the reference's own call pattern (test.py:85-95: the same `IB_lab` / `features_B` objects every frame): the exemplar
        # side (FrameColor.py:20-23 + NonlocalNet.py:452-465,473-476,491-493) is computed on the first call and reused while
        # those tensors, the WarpNet parameters and the kernel selection are unchanged — bit-identical to recomputing it
        bf16 = nonlocal_net._use_bf16(temperature, 1)

        def exemplar_side():
            nB = ops.channel_l2norm_multi((B_relu2_1, B_relu3_1, B_relu4_1, B_relu5_1))
            return nonlocal_net.exemplar_side(IB_lab.detach().contiguous().float(), *nB, bf16=bf16)

        exemplar_cache = nonlocal_net._memo_exemplar_side((IB_lab, B_relu2_1, B_relu3_1, B_relu4_1, B_relu5_1),
                                                          ("warp_color", bf16), exemplar_side)
    if exemplar_cache is None:
        nB = ops.channel_l2norm_multi((B_relu2_1, B_relu3_1, B_relu4_1, B_relu5_1))
        kw = dict(defer_merge=True) if defer_merge else {}
        nonlocal_BA_lab, similarity_map = nonlocal_net(IB_lab, *nA, *nB, temperature=temperature, **kw)
    else:
        # exemplar side cached: the B feature arguments are not read (pass the A ones as placeholders)
        kw = dict(defer_merge=True) if defer_merge else {}
        nonlocal_BA_lab, similarity_map = nonlocal_net(IB_lab, *nA, *nA, temperature=temperature,
                                                       exemplar_cache=exemplar_cache, **kw)
    return nonlocal_BA_lab, similarity_map, features_A


def frame_colorization(IA_lab, IB_lab, IA_last_lab, features_B, vggnet, nonlocal_net, colornet,
                       joint_training=True, feature_noise=0, luminance_noise=0, temperature=0.01,
                       exemplar_cache=None):
    """models/FrameColor.py:41-67.  Returns (IA_ab_predict, nonlocal_BA_lab, features_A_gray).

    `joint_training` only toggles autograd in the reference; this implementation is inference-only
    and its outputs never carry autograd history."""
    IA_lab = IA_lab.detach().contiguous().float()
    IA_l = IA_lab[:, 0:1, :, :]
    if luminance_noise:
        IA_l = IA_l + torch.randn_like(IA_l, requires_grad=False) * luminance_noise
        IA_lab_in = torch.cat((IA_l, IA_lab[:, 1:3]), dim=1).contiguous()
    else:
        IA_lab_in = IA_lab
    # (the correlation's merge is folded into the launch that builds ColorVidNet's input, which also writes the warped Lab this
    # function returns: ops.pack_color_input(want_warped=True); modules that are not this package's WarpNet get the plain call)
    fold = ops.fold_merge() and hasattr(nonlocal_net, "exemplar_side")
    nonlocal_BA_lab, similarity_map, features_A_gray = warp_color(
        IA_l, IB_lab, features_B, vggnet, nonlocal_net, colornet, feature_noise, temperature=temperature,
        exemplar_cache=exemplar_cache, defer_merge=fold)
    # cat((IA_l, nonlocal_BA_ab, similarity_map, IA_last_lab), dim=1)  (FrameColor.py:63-64)
    color_input, nonlocal_BA_lab = ops.pack_color_input(IA_lab_in, nonlocal_BA_lab, similarity_map,
                                                        IA_last_lab.detach().contiguous().float(), want_warped=True)
    IA_ab_predict = colornet(color_input)
    return IA_ab_predict, nonlocal_BA_lab, features_A_gray


class _FrontSlot:
    """One captured front end (dvc_amd/graph.py): static input frame -> (warped Lab, similarity map) at fixed addresses."""

    def __init__(self, cc, shape, device, stream):
        from .graph import CapturedSequence
        self.IA_in = torch.zeros(shape, device=device, dtype=torch.float32)
        self.consumed = None        # event: the last reader of .warped / .sim has been enqueued up to here

        def front():
            warped, sim, _ = warp_color(self.IA_in[:, 0:1], cc.IB_lab, None, cc.vgg, cc.warp, cc.col, 0,
                                        temperature=cc.temperature, exemplar_cache=cc.ex_cache, defer_merge=ops.fold_merge())
            return warped, sim      # (fp32 correlation with the merge folded into the consumer: (ops.CorrPartials, None))

        self.seq = CapturedSequence(front, stream)
        self.warped, self.sim = self.seq.out


class _ColorChain:
    """The captured ColorVidNet chain: static 7-channel input -> ab prediction at a fixed address."""

    def __init__(self, cc, shape, device, stream):
        from .graph import CapturedSequence
        n, _, H, W = shape
        self.cin = torch.zeros((n, 7, H, W), device=device, dtype=torch.float32)
        self.seq = CapturedSequence(lambda: cc.col(self.cin), stream)
        self.ab = self.seq.out


class ClipColorizer:
    """Per-clip driver around `frame_colorization` (the hot loop of /root/reference/test.py:57-96):
    the exemplar's VGG features and its whole WarpNet side (heads, trunk, phi, pooled Lab) are computed
    once per clip instead of once per frame — identical results, 52.6 GFLOP/frame less work
    (SURVEY.md §3.1 "Redundancy").

    graph=True: the per-frame launch sequences (front end, ColorVidNet chain) are captured once per frame geometry into
    hipGraphs and replayed (dvc_amd/graph.py) — bit-identical results.  The per-frame call `frame()` replays both (2 graph
    launches + 4 small kernels instead of ~120 launches: 0.09 instead of 1.56 ms of host time per frame); the pipelined
    `clip()` replays the look-ahead front ends and keeps launching the ColorVidNet chain on the high-priority stream (see
    `graph_parts`).  Needs the exemplar cache (the captured front end reads the cached exemplar side at fixed addresses;
    `set_exemplar` refreshes those buffers in place for the next clip)."""

    def __init__(self, vggnet, nonlocal_net, colornet, temperature=1e-10, cache_exemplar=True, graph=False, batch_plan=False):
        self.vgg, self.warp, self.col = vggnet, nonlocal_net, colornet
        self.temperature = temperature
        self.cache_exemplar = cache_exemplar
        self.graph = bool(graph)
        # batch_plan=True (serving many clips at once: `clip` takes [B,3,H,W] frames = frame t of B independent clips, each with
        # its own exemplar in `set_exemplar(IB [B,3,H,W])`): the Winograd launches of the front ends and of the chain are planned
        # for the whole batch (ops.batch_plan / DVC_CONV_BATCH_PLAN) instead of per image — more frames/s, results deterministic
        # per B but equal to the single-clip runs only to fp32 rounding of the summation order (off: bit-identical to them)
        self.batch_plan = bool(batch_plan)
        # which sequences the PIPELINED driver replays: "front" (default) = the look-ahead front ends, with the ColorVidNet
        # chain launched kernel by kernel on the high-priority stream.  Measured on the MI355X (profiles/
        # r03_graph_overlap_probe.txt): a replayed graph puts its whole backlog of packets into its hardware queue at once,
        # and a high-priority queue with a backlog starves the low-priority ones — both sequences replayed: 3.44 ms/frame
        # (the front ends only run while the recurrence stream is empty), without stream priorities 2.58; front ends
        # replayed + chain launched eagerly 2.40; everything eager 2.48.  "all" / "color" remain for that probe.
        self.graph_parts = "front"
        self.IB_lab = None
        self.features_B = None
        self.ex_cache = None
        self.n_refs = 1              # R > 1 after set_exemplars(): every frame is colourised against R references at once
        self.last_lab = None         # [L, ab] of the last frame colourised by clip()
        self._side_streams = []
        self._main_stream = None
        self._tail_stream = None
        self._graphs = {}            # (kind, shape, slot, ...) -> _FrontSlot / _ColorChain
        self._weights_fp = None      # (data_ptr, version) of every parameter when the weights were last packed
        self._params = None          # (flat parameter list, first parameter of each network) behind that fingerprint

    def prepare(self):
        """Pack all weights of the three networks on the current stream (idempotent, cheap when warm)."""
        for net in (self.vgg, self.warp, self.col):
            prep = getattr(net, "prepare", None)
            if prep is not None:
                prep()

    def _sync_weights(self, refresh_exemplar=True):
        """Make the packed weights (and with them `nets.pack_epoch()`, which marks captured sequences stale) follow the
        parameters: an in-place `load_state_dict` or a `.cuda()` is only noticed by `_PackCache.get`, i.e. by an eager
        forward or `prepare()` — a replayed graph calls neither.  The fingerprint costs ~140 attribute reads; `prepare()`
        runs only when it moved.  `refresh_exemplar=False`: the caller is about to replace the exemplar cache itself."""
        # (the Parameter objects survive load_state_dict / .cuda() — both write through `param.data` / in place — so the module
        # trees are walked once; a replaced Parameter shows up as a changed data_ptr of a dead object only if somebody keeps
        # assigning new nn.Parameter objects, which the identity check of the first parameters catches)
        params = self._params
        if params is None or any(a is not b for a, b in zip(params[1], (next(net.parameters()) for net in (self.vgg, self.warp, self.col)))):
            plist = [p for net in (self.vgg, self.warp, self.col) for p in net.parameters()]
            params = self._params = (plist, [next(net.parameters()) for net in (self.vgg, self.warp, self.col)])
        fp = [(p.data_ptr(), _version_of(p)) for p in params[0]]
        if fp != self._weights_fp:
            prev, self._weights_fp = self._weights_fp, fp       # (recorded first: set_exemplar below comes back through here)
            self.prepare()
            # the cached exemplar side (phi, pooled Lab) was computed with the VGG19 / WarpNet weights of its time: when THOSE
            # moved, it is recomputed from the exemplar (into the same buffers when captured front ends read them) — a new A
            # side must never meet a stale B side
            n_front = sum(1 for net in (self.vgg, self.warp) for _ in net.parameters())
            # (also when the cache came from another rank, parallel.broadcast_exemplar: every rank holds the exemplar and the
            # same new weights, and recomputing locally gives what rank 0 would broadcast)
            if (refresh_exemplar and prev is not None and fp[:n_front] != prev[:n_front] and self.ex_cache is not None
                    and self.IB_lab is not None):
                n = self.n_refs
                self.set_exemplar(self.IB_lab)
                self.n_refs = n

    def set_exemplar(self, IB_lab):
        """test.py:61-66: Lab -> RGB -> VGG features of the reference image."""
        IB_lab = IB_lab.detach().contiguous().float()
        # (records the fingerprint of the weights this exemplar side is computed with: a load_state_dict between this call and
        # the first clip() / frame() must refresh it)
        self._sync_weights(refresh_exemplar=False)
        self.IB_lab = IB_lab
        self.n_refs = 1                            # (set_exemplars raises it after this call)
        rgb = ops.lab2rgb(IB_lab, l_offset=50.0)   # uncenter_l folded into the kernel
        self.features_B = self.vgg(rgb, VGG_OUT, preprocess=True)
        old = self.ex_cache
        self.ex_cache = None
        if self.cache_exemplar:
            nB = ops.channel_l2norm_multi(self.features_B[1:])
            new = self.warp.exemplar_side(IB_lab, *nB, bf16=self.warp._use_bf16(self.temperature, 1))
            self.ex_cache = self._install_cache(old, new)
        return self.features_B

    def set_exemplars(self, IB_labs):
        """All references of a clip in ONE pass (/root/reference/test.py:169-181 colourises the same clip once per reference
        image: R independent recurrences over the SAME frames).  `IB_labs`: a list of [1,3,H,W] Lab references (or one
        [R,3,H,W] tensor).  After this call `clip` / `clip_rgb` / `colorize_video` take the clip's frames ([1,3,H,W] each)
        and return, per frame, the R predictions as one [R,2,H,W] tensor: per frame ONE VGG19(A) + WarpNet(A) front end (114
        of the 348 GFLOP of a frame, identical for every reference), R fused correlations (theta shared; phi / pooled Lab per
        reference), and the ColorVidNet chain at batch R under a batch-aware launch plan (ops.batch_plan: layers that one
        image under-fills drop their split over input channels).  Front-end results are bit-identical to R single-reference
        runs; ab agrees with them to fp32 rounding of the convolutions' summation order (tests/test_gpu_refs.py states the
        measured numbers) and with the oracle per reference within the north-star 1e-3."""
        IB = torch.cat(list(IB_labs), dim=0) if not isinstance(IB_labs, torch.Tensor) else IB_labs
        if not self.cache_exemplar:
            raise RuntimeError("ClipColorizer.set_exemplars needs cache_exemplar=True (the references' WarpNet sides are cached)")
        feats = self.set_exemplar(IB)            # batch R: every launch planned per image -> each reference's cache is
        self.n_refs = IB.shape[0]                # bit-identical to the one set_exemplar(IB[r]) builds
        return feats

    def _rep(self, x):
        """A [1,...] tensor seen as [R,...] (batch stride 0: nothing is copied) in multi-reference mode."""
        return x.expand(self.n_refs, -1, -1, -1) if self.n_refs > 1 and x.shape[0] == 1 else x

    def _chain(self, cin):
        """The ColorVidNet chain; in multi-reference mode the R recurrences advance in lock step as ONE batch, planned as a
        batch (DVC_CONV_BATCH_PLAN)."""
        if self.n_refs > 1 or (self.batch_plan and cin.shape[0] > 1):
            with ops.batch_plan(True):
                return self.col(cin)
        return self.col(cin)

    def _install_cache(self, old, new):
        """Captured front ends read the exemplar cache at the addresses it had at capture time: a new exemplar of the same
        geometry is copied INTO the existing buffers (all streams drained first: a replay in flight may still read them)."""
        if not self._graphs or old is None:
            self._graphs.clear()
            return new
        flat = lambda c: (list(c[0]) if isinstance(c[0], tuple) else [c[0]]) + [c[1]]   # noqa: E731
        fo, fn = flat(old), flat(new)
        if len(fo) != len(fn) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(fo, fn)):
            self._graphs.clear()
            return new
        torch.cuda.synchronize()
        for a, b in zip(fo, fn):
            a.copy_(b)
        return old

    def exemplar_cache_spec(self, lab_shape):
        """[(shape, dtype)] of the flat tensor list `exemplar_cache_tensors()` yields for an exemplar of
        `lab_shape` — what a rank that did not compute the exemplar side must allocate to receive it
        (parallel.broadcast_exemplar).  fp32 correlation: phi [n,256,P], pooled Lab [n,3,h,w]; bf16 candidate
        filter: phi as ([n,P,256] fp32, [n,P,256] bf16 bit patterns in int16), pooled Lab."""
        n, _, H, W = lab_shape
        h, w = int(H / 4), int(W / 4)
        if self.warp._use_bf16(self.temperature, 1):
            return [((n, h * w, 256), torch.float32), ((n, h * w, 256), torch.int16), ((n, 3, h, w), torch.float32)]
        return [((n, 256, h * w), torch.float32), ((n, 3, h, w), torch.float32)]

    def exemplar_cache_shapes(self, lab_shape):
        """Shapes only (kept for callers of the fp32 layout)."""
        return [s for s, _ in self.exemplar_cache_spec(lab_shape)]

    def exemplar_cache_tensors(self):
        """The exemplar cache as a flat list of contiguous tensors, in `exemplar_cache_spec` order."""
        phi, blab = self.ex_cache
        flat = list(phi) if isinstance(phi, tuple) else [phi]
        return [t.contiguous() for t in flat + [blab]]

    def load_exemplar_cache(self, IB_lab, tensors):
        """Install an exemplar cache received from another rank (inverse of exemplar_cache_tensors)."""
        tensors = list(tensors)
        if all(net is not None for net in (self.vgg, self.warp, self.col)):
            self._sync_weights(refresh_exemplar=False)  # (records the weights' fingerprint: a later reload recomputes the cache locally)
        self.IB_lab = IB_lab
        self.n_refs = 1                 # (one exemplar per image of the batch, as after set_exemplar)
        self.features_B = None          # not needed once the exemplar side is cached
        new = ((tensors[0], tensors[1]), tensors[2]) if len(tensors) == 3 else (tensors[0], tensors[1])
        self.ex_cache = self._install_cache(self.ex_cache, new)

    # ---- captured launch sequences ------------------------------------------------------------------------------
    def _captured(self, kind, shape, device, slot, stream):
        # (everything a captured launch bakes in besides addresses: the frame geometry, the temperature — a kernel argument and
        # the choice of the correlation's instantiation — and the correlation precision)
        key = (kind, tuple(shape), slot, float(self.temperature), getattr(self.warp, "corr_precision", "fp32"), device.index,
               self.n_refs)
        g = self._graphs.get(key)
        if g is None:
            # a superseded sequence keeps its activation arena alive: at most two (geometry, temperature, precision)
            # variants stay per (kind, slot) — enough to alternate between two clip sizes without re-capturing
            same = [k for k in self._graphs if k[0] == kind and k[2] == slot and k[5] == device.index]
            if len(same) >= 2:
                torch.cuda.synchronize()
                for k in same[:-1]:
                    del self._graphs[k]
        if g is None or g.seq.stale():
            if self.ex_cache is None:
                raise RuntimeError("ClipColorizer(graph=True) needs the exemplar cache (cache_exemplar=True + set_exemplar)")
            if g is not None:
                torch.cuda.synchronize()        # a replay of the sequence being replaced may still be in flight
            self.prepare()
            g = (_FrontSlot if kind == "front" else _ColorChain)(self, shape, device, stream)
            self._graphs[key] = g
        return g

    def _ensure_streams(self, lookahead):
        """The recurrence stream (highest priority: the ColorVidNet chain is the critical path) and `lookahead` low-priority
        side streams, created once and in this order."""
        lo_prio, hi_prio = torch.cuda.Stream.priority_range()
        if self._main_stream is None:
            self._main_stream = torch.cuda.Stream(priority=hi_prio)
        if len(self._side_streams) < lookahead:
            self._side_streams += [torch.cuda.Stream(priority=lo_prio) for _ in range(lookahead - len(self._side_streams))]
        return self._main_stream, self._side_streams[:lookahead]

    def _capture_stream(self):
        """Stream capture needs a non-default stream; it borrows a side stream instead of creating one of its own.  Measured
        (bench.py on the MI355X, r03): with the per-frame sequences captured on a dedicated extra stream the pipelined driver
        of the same process ran at 321 (eager launches) / 346 (front ends replayed) frames/s instead of 417 / 424 — the
        three streams no longer overlapped.  Extra streams that merely exist and have run a kernel do not reproduce it, nor does
        GPU_MAX_HW_QUEUES=8 change anything (profiles/r03_hw_queue_probe.txt); what inside the HIP runtime ties a graph to the
        stream it was captured on was not isolated — the driver simply never creates that stream."""
        return self._ensure_streams(2)[1][0]

    def frame(self, IA_lab, IA_last_lab, graph=None):
        """One frame_colorization call (the per-frame API of test.py:85): returns (ab, warped Lab).
        In multi-reference mode (set_exemplars) `IA_lab` is the one frame, `IA_last_lab` the R previous [L, ab] tensors
        [R,3,H,W]; returns (ab [R,2,H,W], warped Lab [R,3,H,W]); launches are issued from Python (`graph` is not used)."""
        self._sync_weights()
        if self.n_refs > 1:     # one frame against the R references: IA_last_lab [R,3,H,W] -> (ab [R,2,H,W], warped [R,3,H,W])
            IA_lab = IA_lab.detach().contiguous().float()
            warped, sim, _ = warp_color(IA_lab[:, 0:1], self.IB_lab, None, self.vgg, self.warp, self.col, 0,
                                        temperature=self.temperature, exemplar_cache=self.ex_cache, defer_merge=ops.fold_merge())
            cin, warped = ops.pack_color_input(self._rep(IA_lab), warped, sim, IA_last_lab.detach().contiguous().float(),
                                               want_warped=True)
            return self._chain(cin), warped
        if (self.graph if graph is None else graph) and not self.batch_plan:      # (batch-planned launches: eager, as in clip())
            return self._frame_graph(IA_lab.detach().contiguous().float(),
                                     dict(IA_last_lab=IA_last_lab.detach().contiguous().float()))
        with ops.batch_plan(self.batch_plan):       # (the plan clip() runs under: the two APIs of one object agree bit for bit)
            ab, nl, _ = frame_colorization(IA_lab, self.IB_lab, IA_last_lab, self.features_B, self.vgg, self.warp,
                                           self.col, joint_training=False, feature_noise=0,
                                           temperature=self.temperature, exemplar_cache=self.ex_cache)
        return ab, nl

    def _frame_graph(self, IA_lab, prev):
        """The per-frame call as two graph replays on the CURRENT stream (no look-ahead is possible through this API: the
        next frame is not known).  The outputs are cloned out of the graphs' arenas (the next call overwrites them)."""
        self._sync_weights()        # (a replay never reaches nets._PackCache: reloaded weights are noticed here)
        front = self._captured("front", IA_lab.shape, IA_lab.device, "seq", self._capture_stream())
        chain = self._captured("color", IA_lab.shape, IA_lab.device, 0, self._capture_stream())
        front.IA_in.copy_(IA_lab)
        front.seq.replay()
        _, warped = ops.pack_color_input(IA_lab, front.warped, front.sim, out=chain.cin, want_warped=True, **prev)
        chain.seq.replay()
        return chain.ab.clone(), (warped.clone() if warped is front.warped else warped)

    def clip(self, frames_lab, frame_propagate=False, last=None, lookahead=2, on_frame=None, front_batch=1, graph=None):
        """Recurrence of test.py:68-96; returns the list of ab predictions.

        Only ColorVidNet(t) consumes frame t-1's prediction (test.py:96 -> FrameColor.py:63-64); the
        front end of a frame (VGG19 + WarpNet + correlation) depends on nothing but the frame and the
        exemplar.  With `lookahead` > 0 the front ends of the next `lookahead` frames run on side HIP
        streams while the current stream runs the ColorVidNet chain, so their workgroups fill the CUs
        the other stream's layer leaves idle (few-tile layers, tail rounds).  Same kernels, same
        per-frame arithmetic and order: the predictions are bit-identical to the sequential loop.
        `front_batch` frames share one set of front-end launches (batch dimension): the library plans every
        launch per image (csrc/conv_mfma.hip: conv2d_images), so a batch of N is bit-identical to N single-image
        calls while the per-launch costs are paid once per batch.  (Measured on the MI355X: the front end of four frames
        takes 24 % less time as one batch on an otherwise idle GPU, but inside this driver the other stream already fills
        those gaps — 384-388 frames/s for every batch size — so the default stays 1; the option matters where the host's
        launch rate is the limit.)
        `graph` (default: the constructor's): replay captured launch sequences instead of issuing the launches — one
        hipGraph per side stream for the front end, one for the ColorVidNet chain; same streams, events and priorities.
        `last` (optional) continues the recurrence from an earlier call.  `on_frame(t, IA_lab, ab)`
        (optional) is called right after frame t's launches have been issued, with the recurrence stream
        current — the hook `clip_rgb` hangs the per-frame tail on.
        The previous frame reaches ColorVidNet as its two parts (luminance plane of the previous input frame, previous ab
        prediction: ops.pack_color_input), so test.py:96's cat((IA_l, ab)) is built once, at the end, for `last_lab`."""
        frames_lab = [f.detach().contiguous().float() for f in frames_lab]
        if not frames_lab:
            return []
        self._sync_weights()
        use_graph = (self.graph if graph is None else bool(graph)) and not self.batch_plan      # (batch-planned launches: eager)
        multi = self.n_refs > 1
        if multi:
            # R references in one pass: the frames are single images, every per-frame result carries R images.  The look-ahead
            # front ends can be replayed (a slot holds one frame and R correlations' partial states); the chain is launched
            # kernel by kernel — the pipelined driver's default split (`graph_parts == "front"`)
            if frame_propagate:
                raise ValueError("ClipColorizer.clip: frame_propagate uses the clip's first frame as THE exemplar; it has no "
                                 "multi-reference form (test.py:50 ignores the reference file in that mode)")
            if any(f.shape[0] != 1 for f in frames_lab):
                raise ValueError("ClipColorizer.clip: multi-reference mode takes [1,3,H,W] frames")
            use_graph, front_batch = use_graph and self.graph_parts == "front" and lookahead > 0 and len(frames_lab) >= 2, 1
        if use_graph and int(front_batch) > 1:
            raise ValueError("ClipColorizer.clip: front_batch > 1 is not available with graph=True (a captured front end "
                             "holds one frame per slot); pass graph=False or front_batch=1")
        if last is None:
            last = self.IB_lab if frame_propagate else torch.zeros_like(self._rep(frames_lab[0]))
        last = last.detach().contiguous().float()
        prev = dict(IA_last_lab=last)          # how the previous frame is handed to pack_color_input
        outs = []
        if lookahead <= 0 or len(frames_lab) < 2:
            for IA_lab in frames_lab:
                if use_graph:
                    ab, _ = self._frame_graph(IA_lab, prev)
                else:
                    IA_l = IA_lab[:, 0:1]
                    with ops.batch_plan(self.batch_plan):
                        warped, sim, _ = warp_color(IA_l, self.IB_lab, self.features_B, self.vgg, self.warp, self.col, 0,
                                                    temperature=self.temperature, exemplar_cache=self.ex_cache,
                                                    defer_merge=ops.fold_merge())
                    ab = self._chain(ops.pack_color_input(self._rep(IA_lab), warped, sim, **prev))
                prev = dict(last_l=self._rep(IA_lab), last_ab=ab)
                outs.append(ab)
                if on_frame is not None:
                    on_frame(len(outs) - 1, IA_lab, ab)
            self.last_lab = torch.cat((self._rep(frames_lab[-1])[:, 0:1], outs[-1]), dim=1)   # test.py:96 (pure data movement)
            return outs
        caller = torch.cuda.current_stream()
        # every lazily packed weight is produced here, on the caller's stream, before the fork: a side
        # stream must never be the one that packs a weight another side stream reads (nets._PackCache)
        self.prepare()
        # the ColorVidNet recurrence is the critical path: highest priority; the front ends fill in
        cur, side = self._ensure_streams(lookahead)
        T = len(frames_lab)
        if use_graph:
            return self._clip_graph(frames_lab, prev, caller, cur, side, on_frame)
        for s in side + [cur]:
            s.wait_stream(caller)       # inputs, weights and the exemplar cache were produced on the caller's stream
        fronts = {}
        nb = max(1, int(front_batch))
        if isinstance(self.ex_cache, tuple) and isinstance(self.ex_cache[0], tuple):
            nb = 1                      # (the bf16 candidate-filter correlation takes one image per call)
        batches = [list(range(i, min(i + nb, T))) for i in range(0, T, nb)]

        def launch_front(bi):
            s = side[bi % lookahead]
            with torch.cuda.stream(s):
                fr = [frames_lab[t] for t in batches[bi]]
                IA_lab = fr[0] if len(fr) == 1 else torch.cat(fr, dim=0)
                with ops.batch_plan(self.batch_plan):
                    warped, sim, _ = warp_color(IA_lab[:, 0:1], self.IB_lab, self.features_B, self.vgg, self.warp,
                                                self.col, 0, temperature=self.temperature,
                                                exemplar_cache=self.ex_cache, defer_merge=ops.fold_merge() and nb == 1)
                ev = torch.cuda.Event()
                ev.record(s)
            fronts[bi] = (IA_lab, warped, sim, ev)

        for bi in range(min(lookahead, len(batches))):
            launch_front(bi)
        for bi, members in enumerate(batches):
            IA_b, warped_b, sim_b, ev = fronts.pop(bi)
            cur.wait_event(ev)
            for x in (IA_b, warped_b, sim_b):
                if x is not None:
                    x.record_stream(cur)    # allocated on a side stream, consumed here
            for j, t in enumerate(members):
                if multi:       # (one frame per set of front-end launches; its warped colours / similarity carry R images)
                    IA_lab, warped, sim = self._rep(IA_b), warped_b, sim_b
                elif nb == 1:
                    IA_lab, warped, sim = IA_b, warped_b, sim_b
                else:
                    IA_lab, warped, sim = IA_b[j:j + 1], warped_b[j:j + 1], sim_b[j:j + 1]
                with torch.cuda.stream(cur):
                    ab = self._chain(ops.pack_color_input(IA_lab, warped, sim, **prev))
                    prev = dict(last_l=self._rep(frames_lab[t]), last_ab=ab)
                    if on_frame is not None:
                        on_frame(t, IA_lab, ab)
                outs.append(ab)
                if j == 0 and bi + lookahead < len(batches):   # (issued after the critical-path launches of the frame)
                    launch_front(bi + lookahead)
        with torch.cuda.stream(cur):
            last = torch.cat((self._rep(frames_lab[-1])[:, 0:1], outs[-1]), dim=1)     # test.py:96, once per call
        caller.wait_stream(cur)
        for x in outs + [last]:
            x.record_stream(caller)     # allocated on the recurrence stream, handed to the caller's stream
        self.last_lab = last
        return outs

    def _clip_graph(self, frames_lab, prev, caller, cur, side, on_frame):
        """clip() with captured launch sequences: per frame, the side stream copies the frame into its slot's static input
        and replays the front-end graph; the recurrence stream packs the 7-channel input from the slot's outputs and runs the
        ColorVidNet chain (launched kernel by kernel by default, see `graph_parts`; replayed, its prediction cloned out of
        the graph's arena, otherwise).  Hazards the eager path leaves to the
        allocator are explicit here: a slot's outputs must have been read (event `consumed`, recorded after the pack) before
        the slot's next replay overwrites them; `cin` and `ab` are reused in stream order on the recurrence stream."""
        T, L = len(frames_lab), len(side)
        shape, dev = frames_lab[0].shape, frames_lab[0].device
        self._sync_weights()
        for s in side + [cur]:
            s.wait_stream(caller)
        eager_front = self.graph_parts == "color"       # (experiments: tools/graph_overlap_probe.py)
        eager_color = self.graph_parts == "front"
        slots = [self._captured("front", shape, dev, i, side[i]) for i in range(L)]
        chain = None if eager_color else self._captured("color", shape, dev, 0, self._capture_stream())
        events = {}

        def launch_front(t):
            slot, s = slots[t % L], side[t % L]
            with torch.cuda.stream(s):
                if slot.consumed is not None:
                    s.wait_event(slot.consumed)
                slot.IA_in.copy_(frames_lab[t])
                if eager_front:
                    w_, s_, _ = warp_color(slot.IA_in[:, 0:1], self.IB_lab, None, self.vgg, self.warp, self.col, 0,
                                           temperature=self.temperature, exemplar_cache=self.ex_cache,
                                           defer_merge=isinstance(slot.warped, ops.CorrPartials))
                    slot.warped.copy_(w_)
                    if s_ is not None:
                        slot.sim.copy_(s_)
                else:
                    slot.seq.replay()
                ev = torch.cuda.Event()
                ev.record(s)
            events[t] = ev

        for t in range(min(L, T)):
            launch_front(t)
        outs = []
        for t in range(T):
            slot = slots[t % L]
            cur.wait_event(events.pop(t))
            with torch.cuda.stream(cur):
                cin = ops.pack_color_input(self._rep(frames_lab[t]), slot.warped, slot.sim, out=None if eager_color else chain.cin,
                                           **prev)
                slot.consumed = torch.cuda.Event()
                slot.consumed.record(cur)
                if eager_color:
                    ab = self._chain(cin)
                else:
                    chain.seq.replay()
                    ab = chain.ab.clone()
                prev = dict(last_l=self._rep(frames_lab[t]), last_ab=ab)
                if on_frame is not None:
                    on_frame(t, frames_lab[t], ab)
            outs.append(ab)
            if t + L < T:
                launch_front(t + L)
        with torch.cuda.stream(cur):
            last = torch.cat((self._rep(frames_lab[-1])[:, 0:1], outs[-1]), dim=1)
        caller.wait_stream(cur)
        for s in side:
            caller.wait_stream(s)
        for x in outs + [last]:
            x.record_stream(caller)
        self.last_lab = last
        return outs

    def clip_rgb(self, frames_lab_large, wls_filter_on=True, lambda_value=500, sigma_color=4, frame_propagate=False,
                 lookahead=2, tail_batch=4, last=None):
        """The device-side part of the whole per-frame loop of test.py:68-116: full-resolution Lab frames
        (what `transform(...)` yields, test.py:70) -> x0.5 bilinear -> frame_colorization recurrence -> x2
        bilinear * 1.25 -> WLS filter -> 8-bit RGB (H x W x 3 uint8 device tensors, one per frame).
        The tails run on their own low-priority stream, off the recurrence's critical path, `tail_batch`
        frames per set of launches (the filter is a latency chain: more frames = more lines in flight)."""
        from . import tail
        frames_lab_large = [f.detach().contiguous().float() for f in frames_lab_large]
        small = [tail.downsample_half(f) for f in frames_lab_large]
        caller = torch.cuda.current_stream()
        if self._tail_stream is None:
            self._tail_stream = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[0])
        ts = self._tail_stream
        ts.wait_stream(caller)
        rgbs = [None] * len(small)
        pending = []

        def flush():
            ev = torch.cuda.Event()
            ev.record()                         # on the stream that produced the newest `ab`
            ts.wait_event(ev)
            R = self.n_refs
            idx = [t for t, _ in pending for _ in range(R)]                        # (R tails per frame in multi-reference mode)
            for _, ab in pending:
                ab.record_stream(ts)
            with torch.cuda.stream(ts):
                out, _ = tail.frames_tail([frames_lab_large[t] for t in idx],
                                          [ab if R == 1 else ab[r:r + 1] for _, ab in pending for r in range(R)],
                                          wls_filter_on, lambda_value, sigma_color)
            if R == 1:
                for t, r in zip(idx, out):
                    rgbs[t] = r
            else:
                for i, (t, _) in enumerate(pending):
                    rgbs[t] = out[i * R:(i + 1) * R]
            pending.clear()

        def on_frame(t, IA_lab, ab):
            pending.append((t, ab))
            if len(pending) >= tail_batch or t == len(small) - 1:
                flush()

        self.clip(small, frame_propagate=frame_propagate, last=last, lookahead=lookahead, on_frame=on_frame)
        caller.wait_stream(ts)
        for x in rgbs:
            for y in (x if isinstance(x, list) else [x]):
                y.record_stream(caller)
        return rgbs

    def colorize_video(self, frames_rgb8, reference_rgb8=None, image_size=(432, 768), wls_filter_on=True,
                       lambda_value=500, sigma_color=4, frame_propagate=False, continue_clip=False):
        """colorize_video of test.py:29-121 without its file I/O: 8-bit RGB device frames (any size) in, 8-bit RGB
        colourised frames (image_size) out.  `image_size` is the size CenterPad produces (twice the network
        resolution; the reference's --image_size is the network resolution and test.py:163 doubles it).
        With frame_propagate=True the first frame is the exemplar and `reference_rgb8` is ignored, exactly as
        test.py:50 ignores `reference_file` in that mode.
        `reference_rgb8` may be a LIST of R reference images: the clip is then colourised against all of them in one pass
        (set_exemplars) and every element of the result is a list of R images."""
        from . import tail
        large = [tail.frame_ingest(f, image_size) for f in frames_rgb8]
        if continue_clip:
            # next batch of frames of the SAME clip: keep the exemplar, continue the recurrence (test.py:96)
            if self.IB_lab is None or self.last_lab is None:
                raise RuntimeError("colorize_video(continue_clip=True) needs a previous call on this clip")
            return self.clip_rgb(large, wls_filter_on, lambda_value, sigma_color, last=self.last_lab)
        if not frame_propagate and reference_rgb8 is None:
            raise ValueError("colorize_video: reference_rgb8 is required unless frame_propagate=True")
        if not frame_propagate and isinstance(reference_rgb8, (list, tuple)):
            self.set_exemplars([tail.downsample_half(tail.frame_ingest(r, image_size)) for r in reference_rgb8])
            return self.clip_rgb(large, wls_filter_on, lambda_value, sigma_color)
        ref_large = large[0] if frame_propagate else tail.frame_ingest(reference_rgb8, image_size)
        self.set_exemplar(tail.downsample_half(ref_large))                      # test.py:57-66
        return self.clip_rgb(large, wls_filter_on, lambda_value, sigma_color, frame_propagate=frame_propagate)
