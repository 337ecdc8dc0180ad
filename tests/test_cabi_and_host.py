"""CPU: the C-ABI library loads and exports every symbol include/dvc_hip.h declares (no compute calls
without a GPU), host-side logic (weight packing, geometry, state_dict contract, loud failures)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(debug=False):
    """Entry points include/dvc_hip.h declares: outside (production) or inside (debug=True) its `#ifdef DVC_DEBUG` block."""
    txt = open(os.path.join(ROOT, "include", "dvc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    dbg = re.findall(r"#ifdef DVC_DEBUG(.*?)#endif", txt, flags=re.S)
    txt = "".join(dbg) if debug else re.sub(r"#ifdef DVC_DEBUG.*?#endif", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dvc_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from dvc_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in dvc_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.dvc_abi_version() == _lib.ABI_VERSION
    assert lib.dvc_corr_workspace_bytes(1, 5184) > 0
    # ... and nothing with C linkage is exported that the header does not declare
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3}
    # -fvisibility=hidden + csrc/exports.map: no C++ internals, no host-side kernel handles, no debug hooks
    assert exported == set(syms), (sorted(exported - set(syms)), sorted(set(syms) - exported))
    assert set(_lib.DEBUG_SIGNATURES) == set(_header_symbols(debug=True)) and not (set(_lib.DEBUG_SIGNATURES) & exported)
    dbg_lib = os.path.join(os.path.dirname(_lib.LIB_PATH), "libdvc_hip_debug.so")
    if os.path.exists(dbg_lib):     # the -DDVC_DEBUG build (tools/ only) adds exactly the four hooks
        out = subprocess.run(["nm", "-D", "--defined-only", dbg_lib], capture_output=True, text=True).stdout
        assert {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3} == set(syms) | set(_lib.DEBUG_SIGNATURES)


def test_argument_validation_without_gpu():
    """Validation errors are reported through the return code + dvc_last_error, before any launch."""
    from dvc_amd import _lib
    lib = _lib.load()
    d = _lib.DvcConvDesc(1, 4, 8, 8, 6, 3, 1, 1, 1, 0, 1, 1, 0, 0.0, 0, -1, 0, 0, 0, 0)   # Cout % 4 != 0
    one = ctypes.c_void_p(16)
    rc = lib.dvc_conv2d(ctypes.byref(d), one, one, None, None, None, None, None, None, one, None, 0, None)
    assert rc != 0 and b"multiple of 4" in lib.dvc_last_error()
    oh, ow = ctypes.c_int32(), ctypes.c_int32()
    d2 = _lib.DvcConvDesc(1, 4, 27, 45, 8, 3, 2, 1, 1, 1, 1, 1, 0, 0.0, 0, -1, 0, 0, 0, 0)
    assert lib.dvc_conv2d_out_hw(ctypes.byref(d2), ctypes.byref(oh), ctypes.byref(ow)) == 0
    assert (oh.value, ow.value) == (14, 23)
    # dvc_conv2d_winograd_pool: dilation 1 only, an output of at least one pooling window, a pooled destination
    dp = _lib.DvcConvDesc(1, 64, 27, 48, 64, 3, 1, 2, 2, 0, 1, 1, 1, 0.0, 0, -1, 0, 0, 0, 0, 0)
    rc = lib.dvc_conv2d_winograd_pool(ctypes.byref(dp), one, one, None, None, one, one, 0, None, 0, None)
    assert rc != 0 and b"dilation 1" in lib.dvc_last_error()
    dp = _lib.DvcConvDesc(1, 64, 1, 48, 64, 3, 1, 1, 1, 0, 1, 1, 1, 0.0, 0, -1, 0, 0, 0, 0, 0)
    rc = lib.dvc_conv2d_winograd_pool(ctypes.byref(dp), one, one, None, None, one, one, 0, None, 0, None)
    assert rc != 0 and b"pooling window" in lib.dvc_last_error()
    dp = _lib.DvcConvDesc(1, 64, 27, 48, 64, 3, 1, 1, 1, 0, 1, 1, 1, 0.0, 0, -1, 0, 0, 0, 0, 0)
    rc = lib.dvc_conv2d_winograd_pool(ctypes.byref(dp), one, one, None, None, one, None, 0, None, 0, None)
    assert rc != 0 and b"null argument" in lib.dvc_last_error()


def test_conv_geometry_matches_torch():
    import torch.nn.functional as F

    from dvc_amd import ops
    for (H, W, ks, s, d, p, up, sub) in [(27, 48, 3, 1, 2, 2, 1, 1), (54, 96, 3, 2, 1, 1, 1, 1),
                                         (13, 24, 3, 1, 1, 1, 2, 1), (27, 45, 3, 1, 1, 1, 1, 2),
                                         (12, 20, 1, 1, 1, 0, 1, 1)]:
        x = torch.zeros(1, 1, H, W)
        if sub == 2:
            x = x[:, :, ::2, ::2]
        if up == 2:
            x = F.interpolate(x, scale_factor=2)
        y = F.conv2d(x, torch.zeros(1, 1, ks, ks), stride=s, padding=p, dilation=d)
        assert ops.conv_out_hw(H, W, ks, s, d, p, up, sub) == tuple(y.shape[2:])


def test_pack_conv_weight_layout():
    from dvc_amd import ops
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).view(2, 3, 3, 3)
    p = ops.pack_conv_weight(w)
    assert p.shape == (3, 9, 2)
    for co in range(2):
        for ci in range(3):
            for t in range(9):
                assert p[ci, t, co] == w[co, ci, t // 3, t % 3]


def test_state_dict_contract_and_loud_cpu_failure(weights, capsys):
    from models.ColorVidNet import ColorVidNet
    from models.NonlocalNet import VGG19_pytorch, WarpNet
    vgg, warp, col = VGG19_pytorch(), WarpNet(1), ColorVidNet(7)
    out = capsys.readouterr().out
    assert "replace all deconv with [nearest + conv]" in out      # ColorVidNet.py:80,85
    for m, sd in zip((vgg, warp, col), weights):
        m.load_state_dict(sd, strict=True)
        assert list(m.state_dict().keys()) == list(sd.keys())
        assert all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
        assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in sd.values())
    assert len(weights[0]) == 32 and len(weights[1]) == 43 and len(weights[2]) == 65
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vgg(torch.zeros(1, 3, 16, 16), ["r11"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        col(torch.zeros(1, 7, 16, 16))
    from utils.util import feature_normalize, uncenter_l
    assert uncenter_l(-50.0) == 0.0
    with pytest.raises(RuntimeError):
        feature_normalize(torch.zeros(1, 4, 2, 2))


def test_product_code_never_imports_oracle():
    """The oracle is test infrastructure; nothing under the product package may import it."""
    pkg = os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                for ln in open(os.path.join(dp, f)).read().splitlines():
                    assert not re.match(r"\s*(from|import)\s+oracle", ln), (f, ln)
                    assert "dvc_oracle" not in ln, (f, ln)


def test_fma_division_by_constant_is_correctly_rounded():
    """csrc/corr.hip computes ATen's s = fl32(f / T) for the fixed divisor T as q = f*y followed by two fma
    residual corrections (y = fl32(1/T)).  Check the recipe against exact rational arithmetic: it must
    give the correctly rounded quotient, otherwise ties at T = 1e-10 would differ from the reference."""
    import math
    from fractions import Fraction

    import numpy as np

    def rn32(x):
        """Round a Fraction to the nearest float32 (ties to even); returns a python float."""
        if x == 0:
            return 0.0
        sign = -1 if x < 0 else 1
        x = abs(x)
        e = math.floor(math.log2(x.numerator) - math.log2(x.denominator))
        while Fraction(2) ** e > x:
            e -= 1
        while Fraction(2) ** (e + 1) <= x:
            e += 1
        e = max(e, -126)
        ulp = Fraction(2) ** (e - 23)
        q = x / ulp
        n = q.numerator // q.denominator
        rem = q - n
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
            n += 1
        return float(sign * n * ulp)

    def fma(a, b, c):
        return rn32(Fraction(a) * Fraction(b) + Fraction(c))

    rng = np.random.default_rng(0)
    for T in (1e-10, 0.01, 0.005, 1.0, 3.3e-7, 1e-4, 0.7):
        T = float(np.float32(T))
        y = rn32(Fraction(1) / Fraction(T))
        fs = np.concatenate([rng.uniform(-1, 1, 160), rng.uniform(0.2, 0.3, 60), rng.uniform(-1e-3, 1e-3, 20),
                             [1.0, -1.0, 0.0, 0.5, 2.0 ** -20, 1 - 2.0 ** -24]]).astype(np.float32)
        for f in fs:
            f = float(f)
            q = rn32(Fraction(f) * Fraction(y))
            q = fma(fma(-T, q, f), y, q)
            q = fma(fma(-T, q, f), y, q)
            assert q == rn32(Fraction(f) / Fraction(T)), (T, f)


def test_winograd_plan_is_a_pure_function_of_the_layer_not_of_the_batch():
    """Host logic, no GPU: the split over input channels the Winograd launcher would use (dvc_conv2d_winograd_split) and the
    engine choice (ops.winograd_selected) must not depend on the batch size — the clip driver's batched front ends and the
    bit-identical-batch tests rest on it — and the split is one the kernel supports."""
    import ctypes
    from dvc_amd import _lib, ops
    lib = _lib.load()
    ws = 64 << 20
    layers = [(256, 256, 54, 96, 1, 1), (512, 512, 27, 48, 1, 1), (512, 512, 27, 48, 2, 1), (128, 128, 216, 384, 1, 1),
              (64, 64, 216, 384, 1, 1), (512, 512, 13, 24, 1, 1), (256, 64, 13, 24, 1, 2), (128, 256, 54, 96, 1, 1)]
    for (ci, co, H, W, dil, up) in layers:
        got = []
        for N in (1, 2, 5):
            d = _lib.DvcConvDesc(N, ci, H, W, co, 3, 1, dil, dil, 0, up, 1, 1, 0.0, 0, -1, 0, 0, 0, 0, 0)
            sp, ipl = ctypes.c_int32(0), ctypes.c_int32(0)
            assert lib.dvc_conv2d_winograd_split(ctypes.byref(d), ws, ctypes.byref(sp), ctypes.byref(ipl)) == 0, lib.dvc_last_error()
            got.append(sp.value)
            # images one launch covers: the whole batch unless the workspace (partial sums of the split) or the
            # 65535-workgroup cap says otherwise — then the host must not defer the reduce (ops.conv2d_winograd)
            cap = ws // (sp.value * co * (H * up) * (W * up) * 4) if sp.value > 1 else N
            assert 1 <= ipl.value <= N and ipl.value <= max(cap, 1)
            assert ops.winograd_selected(N, ci, H, W, co, dil=dil, pad=dil, in_up=up) == \
                ops.winograd_selected(1, ci, H, W, co, dil=dil, pad=dil, in_up=up)
        assert got[0] == got[1] == got[2] and 1 <= got[0] <= 8, (ci, co, H, W, got)
    # without a workspace there is nothing to split into
    d = _lib.DvcConvDesc(1, 256, 54, 96, 256, 3, 1, 1, 1, 0, 1, 1, 1, 0.0, 0, -1, 0, 0, 0, 0, 0)
    sp, ipl = ctypes.c_int32(0), ctypes.c_int32(0)
    assert lib.dvc_conv2d_winograd_split(ctypes.byref(d), 0, ctypes.byref(sp), ctypes.byref(ipl)) == 0 and sp.value == 1
    # a batch whose partial sums do not fit the workspace is covered in several launches: images_per_launch < N
    big = _lib.DvcConvDesc(64, 64, 216, 384, 64, 3, 1, 1, 1, 0, 1, 1, 1, 0.0, 0, -1, 2, 0, 0, 0, 0)
    assert lib.dvc_conv2d_winograd_split(ctypes.byref(big), ws, ctypes.byref(sp), ctypes.byref(ipl)) == 0
    assert sp.value == 2 and ipl.value == ws // (2 * 64 * 216 * 384 * 4) < 64
    # a layer the kernel does not take is refused with a message, not planned
    bad = _lib.DvcConvDesc(1, 3, 54, 96, 64, 3, 1, 1, 1, 0, 1, 1, 1, 0.0, 0, -1, 0, 0, 0, 0, 0)
    assert lib.dvc_conv2d_winograd_split(ctypes.byref(bad), ws, ctypes.byref(sp), ctypes.byref(ipl)) != 0
    assert b"Cin" in lib.dvc_last_error()


def test_batch_plan_flag_trades_the_split_for_the_batch():
    """DVC_CONV_BATCH_PLAN (include/dvc_hip.h): the plan of the WHOLE batch — the multi-reference pass runs R ColorVidNet
    recurrences in lock step, so the R images' workgroups fill the chip together and an under-filled layer needs a smaller
    split over input channels (or none).  Host logic: the flag never changes a single image's plan, never raises the split,
    lowers it on the network's under-filled layers at R = 4, and `ops.batch_plan` sets it only while active."""
    import ctypes
    from dvc_amd import _lib, ops
    lib = _lib.load()
    ws = 64 << 20

    def split(N, ci, co, H, W, dil, flags):
        d = _lib.DvcConvDesc(N, ci, H, W, co, 3, 1, dil, dil, 0, 1, 1, 1, 0.0, 0, -1, 0, 0, 0, 0, flags)
        sp, ipl = ctypes.c_int32(0), ctypes.c_int32(0)
        assert lib.dvc_conv2d_winograd_split(ctypes.byref(d), ws, ctypes.byref(sp), ctypes.byref(ipl)) == 0, lib.dvc_last_error()
        return sp.value

    lowered = 0
    for (ci, co, H, W, dil) in [(256, 256, 54, 96, 1), (512, 512, 27, 48, 1), (512, 512, 27, 48, 2), (128, 128, 216, 384, 1),
                                (256, 256, 108, 192, 1), (128, 256, 54, 96, 1)]:
        base = split(1, ci, co, H, W, dil, 0)
        assert split(1, ci, co, H, W, dil, ops.BATCH_PLAN) == base
        assert split(4, ci, co, H, W, dil, 0) == base
        s4 = split(4, ci, co, H, W, dil, ops.BATCH_PLAN)
        assert 1 <= s4 <= base, (ci, co, H, W, base, s4)
        lowered += s4 < base
    assert lowered >= 2
    assert ops._plan_flags(4) == 0
    with ops.batch_plan(True):
        assert ops._plan_flags(4) == ops.BATCH_PLAN and ops._plan_flags(1) == 0
        with ops.batch_plan(False):
            assert ops._plan_flags(4) == 0
    assert not ops.batch_plan_enabled()


def test_bench_torchrun_relaunch_command():
    """`python bench.py --gpus N` re-launches itself as the driver does (one rank per GPU under torch.distributed.run, 127.0.0.1
    rendezvous, its own flags passed through): the command line is built here without launching anything."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd, env = bench.torchrun_command(8, argv, 29511, environ={"PATH": "/usr/bin"})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == argv                       # the ranks get exactly the caller's flags (--gpus N included)
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin"
    _, env1 = bench.torchrun_command(2, [], 1, environ={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert env1["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"      # an explicit setting of the caller is kept


def test_error_aware_engine_map_host_logic():
    """ops.winograd_selected with named layers: under `auto` the layers of the engine map stay on the direct engine, `speed` is
    the geometry rule alone, `winograd` / `direct` force one engine; the map never depends on the batch size; every name in the
    built-in map is a real 3x3 layer of one of the three networks."""
    from dvc_amd import arch, ops
    old_algo, old_map = ops.conv_algo(), ops._direct_layers
    try:
        geo = dict(dil=1, pad=1)
        ops.set_conv_algo("auto")
        ops.set_direct_layers(["cvn.conv2_2"])
        assert not ops.winograd_selected(1, 128, 108, 192, 128, layer="cvn.conv2_2", **geo)
        assert not ops.winograd_selected(4, 128, 108, 192, 128, layer="cvn.conv2_2", **geo)
        assert ops.winograd_selected(1, 128, 108, 192, 128, layer="cvn.conv9_2", **geo)
        assert ops.winograd_selected(1, 128, 108, 192, 128, **geo)                  # unnamed calls: geometry rule
        ops.set_conv_algo("speed")
        assert ops.winograd_selected(1, 128, 108, 192, 128, layer="cvn.conv2_2", **geo)
        ops.set_conv_algo("winograd")
        assert ops.winograd_selected(1, 128, 108, 192, 128, layer="cvn.conv2_2", **geo)
        ops.set_conv_algo("direct")
        assert not ops.winograd_selected(1, 128, 108, 192, 128, layer="cvn.conv9_2", **geo)
        ops.set_direct_layers(None)
        assert ops.direct_layers() == arch.DIRECT_LAYERS
        with pytest.raises(ValueError):
            ops.set_conv_algo("fastest")
    finally:
        ops.set_conv_algo(old_algo)
        ops._direct_layers = old_map
    names = {"vgg." + n for n, _, _ in arch.VGG_CONVS}
    names |= {f"warp.{h}.{ci}" for h in arch.WARP_HEAD_ORDER for (ci, _, _, _, _) in arch.WARP_HEADS[h]["convs"]}
    names |= {f"warp.layer.{b}.conv{k}" for b in range(arch.WARP_NUM_RESBLOCKS) for k in (1, 2)}
    names |= {"cvn." + c["key"] for c in arch.CVN_CONVS}
    assert arch.DIRECT_LAYERS <= names, sorted(arch.DIRECT_LAYERS - names)


def test_exemplar_memo_keys_on_identity_versions_and_weights():
    """nets.WarpNet._memo_exemplar_side (the cache behind the reference's unmodified call pattern), on CPU tensors with a
    counting stand-in for the exemplar side: hit on the same tensor objects, miss on new objects, on an in-place write
    (version counter), on a parameter update, on another regime / engine choice; off switch."""
    import contextlib
    import io
    from dvc_amd import ops
    from dvc_amd.nets import WarpNet
    with contextlib.redirect_stdout(io.StringIO()):
        net = WarpNet(1)
    calls = []

    def compute():
        calls.append(1)
        return ("phi%d" % len(calls), "blab")

    a, b = torch.zeros(3), torch.zeros(4)
    v1 = net._memo_exemplar_side((a, b), ("warp_color", False), compute)
    assert net._memo_exemplar_side((a, b), ("warp_color", False), compute) is v1 and len(calls) == 1
    assert net._memo_exemplar_side((a.clone(), b), ("warp_color", False), compute) is not v1 and len(calls) == 2     # new object
    net._memo_exemplar_side((a, b), ("warp_color", False), compute)
    n = len(calls)
    a.add_(1.0)                                                                                                      # in-place write
    net._memo_exemplar_side((a, b), ("warp_color", False), compute)
    assert len(calls) == n + 1
    net._memo_exemplar_side((a, b), ("warp_color", True), compute)                                                   # other regime
    assert len(calls) == n + 2
    with torch.no_grad():
        net.theta.weight.mul_(2.0)                                                                                    # parameter update
    net._memo_exemplar_side((a, b), ("warp_color", True), compute)
    assert len(calls) == n + 3
    old = ops.conv_algo()
    try:
        ops.set_conv_algo("direct")
        net._memo_exemplar_side((a, b), ("warp_color", True), compute)
        assert len(calls) == n + 4
    finally:
        ops.set_conv_algo(old)
    ops.set_exemplar_memo(False)
    try:
        net._memo_exemplar_side((a, b), ("warp_color", True), compute)
        net._memo_exemplar_side((a, b), ("warp_color", True), compute)
        assert len(calls) == n + 6
    finally:
        ops.set_exemplar_memo(True)
    # the memo is not part of the module's state
    assert not any("memo" in k for k in net.state_dict())
    # r06: tensors without a version counter (torch.inference_mode) bypass the memo instead of raising
    with torch.inference_mode():
        ai = torch.zeros(3)
    n = len(calls)
    net._memo_exemplar_side((ai, b), ("warp_color", True), compute)
    net._memo_exemplar_side((ai, b), ("warp_color", True), compute)
    assert len(calls) == n + 2
    # "verify": every hit recomputes and compares; a stand-in whose value changes between calls is reported and replaced
    import warnings
    ops.set_exemplar_memo("verify")
    try:
        state = {"v": torch.ones(2)}
        comp = lambda: (calls.append(1), (state["v"].clone(), "blab"))[1]        # noqa: E731
        net._memo_exemplar_side((a, b), ("warp_color", False), comp)
        n = len(calls)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            r = net._memo_exemplar_side((a, b), ("warp_color", False), comp)
        assert len(calls) == n + 1 and not w and torch.equal(r[0], torch.ones(2))
        state["v"] = torch.full((2,), 2.0)          # "the exemplar changed through .data": same identities, same versions
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            r = net._memo_exemplar_side((a, b), ("warp_color", False), comp)
        assert torch.equal(r[0], torch.full((2,), 2.0)) and any(issubclass(x.category, RuntimeWarning) for x in w)
    finally:
        ops.set_exemplar_memo(True)
    assert ops.exemplar_memo_mode() == "on"


def test_training_side_gemm_guard_refuses_reduced_precision_and_cpu_tensors():
    """ops.bmm (r06: the training side's plain batched GEMMs on the vendor library) computes in plain fp32 or not at all: a
    relaxed float32 matmul precision raises (with the way out in the message), and so do CPU tensors — no silent fallback."""
    import torch
    from dvc_amd import ops
    before = torch.get_float32_matmul_precision()
    try:
        torch.set_float32_matmul_precision("high")
        with pytest.raises(RuntimeError, match="DVC_GEMM_LIB=0"):
            ops.bmm(torch.zeros(1, 2, 2), torch.zeros(1, 2, 2))
    finally:
        torch.set_float32_matmul_precision(before)
    with pytest.raises(RuntimeError, match="ROCm tensor"):
        ops.bmm(torch.zeros(1, 2, 2), torch.zeros(1, 2, 2))
    assert ops.gemm_lib() in (True, False)
