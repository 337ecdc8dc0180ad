#!/usr/bin/env python
"""Benchmark of the hot path: colourised 216x384 frames/s/GPU (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one frame of the clip recurrence of /root/reference/test.py:68-96 (frame t consumes frame t-1's
prediction): VGG19(A) -> WarpNet(A) + fused correlation -> ColorVidNet on one synthetic 1x3x216x384 Lab frame
already resident in HBM.  The K timed steps go through `ClipColorizer.clip` (front end of the next frames on side
HIP streams, ColorVidNet recurrence on the main one; bit-identical to per-frame `frame_colorization` calls, which
are timed right after and reported as `config.per_frame_api_frames_per_s`; `--lookahead 0` times those instead).
The K-step clip is timed `config.repeats` times back to back (same frames, same recurrence start, each repeat bracketed by
barrier + synchronize; enough repeats for >= 0.5 s of timed GPU work whatever K is): `ms_per_step` / `value` are the MEDIAN
repeat, p10 / p90 are in `config` (">= 50 timed, median + p10/p90").
N=1 runs BASELINE.json configs[1].  With N>1 each rank colourises its own contiguous chunk of K frames (weak
scaling; the exemplar-side tensors are computed on rank 0 and broadcast once over RCCL/xGMI; no collective in
the per-frame path).  `--gpus N` with N > 1 outside torch.distributed.run re-launches itself under it (one process
per GPU); a world size different from --gpus, or fewer visible GPUs than requested, is an error — the line never
reports an `n_gpus` other than the one asked for.  Rank 0 prints ONE JSON line on stdout; diagnostics go to stderr.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H, W = 216, 384
C, P = 256, (H // 4) * (W // 4)
# algorithmic work of the correlation stage per frame (= per launch)
CORR_FLOPS = 2.0 * P * P * C + 2.0 * P * P * 3          # 13.92 GFLOP
CORR_BYTES = 4.0 * (2 * P * C + 3 * P + 3 * P + P)      # 10.76 MB compulsory traffic
PATH_FLOPS = 348.4e9                                     # minimal whole-path FLOPs / frame
PEAK_F32_MFMA_TFLOPS = 157.3                             # MI355X_MICROARCH.md, fp32 matrix
PEAK_BF16_MFMA_TFLOPS = 2500.0                           # dense bf16 matrix peak (never the 2:1-sparsity figure)
PEAK_HBM_GBS = 8000.0
# HBM traffic of one corr_fwd_kernel launch at P=5184.  rocprofv3 --pmc cannot run inside this process, so this
# is an OFFLINE measurement (tools/pmc_corr.sh -> tools/summarize_profiles.py), not a number of this run: the
# line names the summary file it was read from and that file's hash.  (2 x FETCH_SIZE + WRITE_SIZE, the gfx950
# correction of MI355X_MICROARCH.md; phi is re-streamed through the eight per-XCD L2s, hence >> 10.76 MB.)
CORR_TRAFFIC_FILE = "profiles/corr_traffic.json"


def corr_traffic():
    """(bytes per launch or None, provenance dict) read from the committed offline PMC summary."""
    import hashlib
    path = os.path.join(ROOT, CORR_TRAFFIC_FILE)
    try:
        raw = open(path, "rb").read()
        rec = json.loads(raw)
        src = hashlib.sha256(open(os.path.join(PKG, "csrc", "corr.hip"), "rb").read()).hexdigest()[:16]
        if rec.get("corr_hip_sha256_16") != src:
            # the PMC passes ran another version of the kernel's source: a constant that silently goes stale is worse than none
            return None, {"kind": "stale: csrc/corr.hip (sha256 %s) is not the source the PMC passes measured (%s); re-run "
                                  "tools/gpu_final.sh + tools/summarize_profiles.py" % (src, rec.get("corr_hip_sha256_16")),
                          "file": CORR_TRAFFIC_FILE}
        return float(rec["bytes_per_launch"]), {"kind": "offline rocprofv3 --pmc passes, not re-measured by this run",
                                                "file": CORR_TRAFFIC_FILE, "sha256_16": hashlib.sha256(raw).hexdigest()[:16],
                                                "measured_on": rec.get("measured_on"), "P": rec.get("P")}
    except (OSError, ValueError, KeyError):
        return None, {"kind": "unavailable", "file": CORR_TRAFFIC_FILE}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_nets(device):
    from dvc_amd import synth
    from models.ColorVidNet import ColorVidNet
    from models.NonlocalNet import VGG19_pytorch, WarpNet
    sd = (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))
    with contextlib.redirect_stdout(io.StringIO()):
        nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
    for m, s in zip(nets, sd):
        m.load_state_dict(s)
        m.eval().to(device)
    return nets, sd


def host_cpus():
    """(usable hardware threads, how that was determined): the affinity mask, capped by the cgroup CPU quota — a
    container may see 256 threads in its mask while being allowed 32 CPUs' worth of time, and an ATen thread pool
    sized for the mask then runs ~100x slower than one sized for the quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    how = "sched_getaffinity"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period))
                if q < n:
                    n, how = q, f"cgroup quota ({path})"
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, how


def _cpu_leg(sd, threads, n_warm, n_timed, budget_s):
    """Median s/frame of the oracle's frame_colorization recurrence on `threads` host threads; gives up (returning
    what it has) once `budget_s` seconds are spent — the leg must never eat the GPU box's time."""
    from dvc_amd import synth
    from oracle import dvc_oracle as O
    torch.set_num_threads(threads)
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, 216, 384)
    t_start = time.perf_counter()
    with torch.no_grad():
        fB = O.exemplar_features(IB, sd[0])                # once per clip, excluded like on the GPU side
        last = torch.zeros(1, 3, 216, 384)
        times, warm = [], []
        for i in range(n_warm + n_timed):
            fr = synth.synth_lab(synth.FRAME_SEED0 + i, 216, 384)
            t0 = time.perf_counter()
            ab, _, _ = O.frame_colorization(fr, IB, last, fB, *sd, temperature=1e-10)
            dt = time.perf_counter() - t0
            last = torch.cat((fr[:, 0:1], ab), 1)
            (times if i >= n_warm else warm).append(dt)
            if time.perf_counter() - t_start > budget_s:
                break
    done = sorted(times) if times else sorted(warm)
    return done[len(done) // 2], len(times)


def cpu_baseline(sd):
    """the reference's path on this box's host cores, same synthetic clip, fp32, flush-denormal,
    k = every usable hardware thread and k = 1, 2 warm-up + 5 timed frames each, median.  What is timed is the oracle — the
    op-for-op torch-CPU restatement that oracle/pin_reference.py shows bit-identical to the unmodified reference
    modules (`kind: "port"`; /root/reference does not exist on the GPU box).  ATen's CPU kernels stop scaling well
    below a 128-thread host, so a 32-thread leg is timed too and `value` is the FASTEST leg.  Every leg has a time
    budget (the r02 box advertised 256 threads under a smaller CPU quota: 77 s per frame with a 256-thread pool)."""
    avail, how = host_cpus()
    torch.set_flush_denormal(True)
    keep = torch.get_num_threads()
    legs = {}

    def probe(k):   # one 6-GFLOP convolution, best of 3: a cheap look at how a k-thread pool behaves on this host
        torch.set_num_threads(k)
        x, w = torch.randn(1, 64, 216, 384), torch.randn(64, 64, 3, 3)
        best = float("inf")
        with torch.no_grad():
            for _ in range(3):
                t0 = time.perf_counter()
                torch.nn.functional.conv2d(x, w, padding=1)
                best = min(best, time.perf_counter() - t0)
        return best

    try:
        # (2 warm-up + >= 5 timed frames per leg; the one-thread leg costs ~1.8 s per frame)
        for k, warm, timed, budget in sorted({(avail, 2, 5, 20.0), (min(avail, 32), 2, 5, 20.0), (1, 2, 5, 30.0)}):
            if k > 32:
                p_all, p_32 = probe(k), probe(32)
                if p_all > 3.0 * p_32:
                    legs[k] = {"frames_per_s": None, "note": f"skipped: a {k}-thread ATen pool is oversubscribed on this host "
                                                             f"(6-GFLOP conv probe {p_all * 1e3:.0f} ms vs {p_32 * 1e3:.0f} ms on 32 threads)"}
                    log(f"[bench] cpu baseline, {k} threads: skipped ({legs[k]['note']})")
                    continue
            med, n_done = _cpu_leg(sd, k, warm, timed, budget)
            legs[k] = {"frames_per_s": round(1.0 / med, 4), "median_ms_per_frame": round(med * 1e3, 1),
                       "warmup_frames": warm, "timed_frames": n_done}
            if n_done < timed:
                legs[k]["note"] = f"stopped after {budget:.0f} s"
            log(f"[bench] cpu baseline, {k} thread(s): {med * 1e3:.0f} ms/frame ({n_done} timed)")
    finally:
        torch.set_num_threads(keep)
    best = max((k for k in legs if legs[k]["frames_per_s"]), key=lambda k: legs[k]["frames_per_s"])
    cpu_model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": legs[best]["frames_per_s"], "unit": "frames/s", "cores": best, "kind": "port",
            "sample": "216x384 frames of the same synthetic clip (exemplar seed 2, frames 1000..), oracle "
                      "frame_colorization recurrence = the reference op for op (exemplar side recomputed per frame as "
                      "the reference does), torch CPU fp32, flush-denormal; 2 warm-up + 5 timed frames per leg, median; "
                      f"`value` is the fastest leg ({best} threads)",
            "host": {"threads_usable": avail, "threads_usable_from": how, "cpu": cpu_model},
            "by_threads": {str(k): v for k, v in sorted(legs.items())}}


def parity_block(cc, sd, device, n_pool=6):
    """Untimed: how far the TIMED engine is from the fp64 truth at the timed configuration, next to the reference-equivalent
    CPU fp32 run against the same truth (SURVEY.md 7 hard part 1 / 8c: "must be <= the CPU figure").  216x384 frames as first
    frames of a clip (exemplar seed 2, frame seeds 1000.., the plain seed-0 weights `value` is measured with), T = 1e-10; the
    oracle runs on the host CPU in fp32 and fp64 (checker only).
    r06: the top-level ratios are POOLED over `n_pool` frames (frames on which an arg-max differs from the truth's are left out on
    both sides); `frame_1000` is what r05 reported (that frame alone) and `cpu32_other_thread_count` calibrates it — the SAME
    reference arithmetic at another ATen thread count against the same truth: with the chaotic random weights the tail of one
    frame's error field is a lottery over rounding noise, the reference's own q99.9 / max move by 1.5x / 2x between thread
    counts on that frame.  tests/test_gpu_nets.py asserts the pooled and the single-frame comparison."""
    import numpy as np
    from dvc_amd import ops, synth
    from oracle import dvc_oracle as O
    keep = torch.get_num_threads()
    k_main = max(1, min(32, host_cpus()[0]))
    k_other = 4 if k_main != 4 else 2
    torch.set_num_threads(k_main)
    torch.set_flush_denormal(True)
    try:
        IB = synth.synth_lab(synth.EXEMPLAR_SEED, 216, 384)
        sd64 = tuple(O.to_dtype(s, torch.float64) for s in sd)
        st = lambda e: {"max": float(e.max()), "q999": float(np.quantile(e.numpy().ravel(), 0.999)), "mean": float(e.mean()),   # noqa: E731
                        "rms": float((e ** 2).mean().sqrt())}
        sig = lambda d: {k: float("%.4g" % v) for k, v in d.items()}        # noqa: E731
        rat = lambda a, b: {k: round(a[k] / b[k], 3) for k in a}            # noqa: E731
        t0 = time.perf_counter()
        with torch.no_grad():
            fB32, fB64 = O.exemplar_features(IB, sd[0]), O.exemplar_features(IB.double(), sd64[0])
        eg_all, ec_all, per_frame, first = [], [], [], None
        for i in range(max(1, n_pool)):
            fr = synth.synth_lab(synth.FRAME_SEED0 + i, 216, 384)
            z = torch.zeros_like(fr)
            with torch.no_grad():
                ab32, w32, _ = O.frame_colorization(fr, IB, z, fB32, *sd, temperature=1e-10)
                ab64, w64, _ = O.frame_colorization(fr.double(), IB.double(), z.double(), fB64, *sd64, temperature=1e-10)
            ab, wl = cc.frame(fr.to(device), z.to(device), graph=False)
            eg, ec = (ab.double().cpu() - ab64).abs(), (ab32.double() - ab64).abs()
            flip = (wl.double().cpu() - w64).abs().max().item() > 1e-3 or (w32.double() - w64).abs().max().item() > 1e-3
            if i == 0:
                torch.set_num_threads(k_other)
                try:
                    with torch.no_grad():
                        ab32o = O.frame_colorization(fr, IB, z, O.exemplar_features(IB, sd[0]), *sd, temperature=1e-10)[0]
                finally:
                    torch.set_num_threads(k_main)
                first = (fr, z, ab64, st(eg), st(ec), st((ab32o.double() - ab64).abs()))
            per_frame.append({"seed": synth.FRAME_SEED0 + i, "argmax_flip": bool(flip), "gpu_over_cpu32": rat(st(eg), st(ec))})
            if not flip:
                eg_all.append(eg.flatten())
                ec_all.append(ec.flatten())
        t_cpu = time.perf_counter() - t0
        fr, z, ab64, g, c, co = first
        pooled = None
        if eg_all:
            G, Cc = st(torch.cat(eg_all)), st(torch.cat(ec_all))
            pooled = {"frames": len(eg_all), "gpu_vs_fp64": sig(G), "cpu32_vs_fp64": sig(Cc), "gpu_over_cpu32": rat(G, Cc), "per_frame": per_frame}
        speed = None
        if ops.conv_algo() == "auto":
            # what the geometry-only Winograd rule (r01-r04's engine choice, `config.engine_speed`) would have cost here
            try:
                ops.set_conv_algo("speed")
                ab_s, _ = cc.frame(fr.to(device), z.to(device), graph=False)
                gs = st((ab_s.double().cpu() - ab64).abs())
                speed = {"gpu_vs_fp64": sig(gs), "gpu_over_cpu32": rat(gs, c)}
            finally:
                ops.set_conv_algo("auto")
        top = pooled if pooled is not None else {"gpu_vs_fp64": sig(g), "cpu32_vs_fp64": sig(c), "gpu_over_cpu32": rat(g, c), "frames": 1}
        return {"gpu_vs_fp64": top["gpu_vs_fp64"], "cpu32_vs_fp64": top["cpu32_vs_fp64"], "gpu_over_cpu32": top["gpu_over_cpu32"],
                "frames_pooled": top["frames"], "per_frame": per_frame,
                "frame_1000": {"gpu_vs_fp64": sig(g), "cpu32_vs_fp64": sig(c), "gpu_over_cpu32": rat(g, c)},
                "cpu32_other_thread_count": {"threads": k_other, "frame_1000_vs_fp64": sig(co), "over_cpu32": rat(co, c), "gpu_over_it": rat(g, co),
                                             "note": f"the reference arithmetic itself (oracle, torch CPU fp32) at {k_other} instead of "
                                                     f"{k_main} threads against the same fp64 truth on frame 1000: how much of a single "
                                                     "frame's tail ratio is rounding-noise lottery (profiles/r06_parity_hotspot.txt, "
                                                     "r06_parity_pool_probe.txt)"},
                "conv_algo": ops.conv_algo(), "direct_layers": sorted(ops.direct_layers()) if ops.conv_algo() == "auto" else None,
                "engine_speed_for_comparison": speed,
                "sample": "ab of 216x384 frames as first frames of a clip (exemplar seed 2, frame seeds 1000.., plain seed-0 weights, "
                          "T = 1e-10): |GPU fp32 - oracle fp64| and |oracle fp32 (= the reference's CPU run) - oracle fp64| over the "
                          f"2 x 216 x 384 values per frame.  Top level = POOLED over the {top['frames']} frames without an arg-max "
                          "flip (r05 reported frame 1000 alone: `frame_1000`; one frame's q99.9 / max are a lottery over rounding "
                          "noise — the reference's own run moves by `cpu32_other_thread_count.over_cpu32` between thread counts); "
                          f"engine_speed_for_comparison: frame 1000; untimed, oracle on the host CPU ({t_cpu:.0f} s)"}
    finally:
        torch.set_num_threads(keep)


def executed_matrix_flops(cc, frame, last):
    """(executed, direct-equivalent) matrix FLOPs of ONE frame as the HIP path runs it: every convolution launch of a per-frame
    call is recorded (ops.conv_record) and priced by the engine that ran it — direct implicit GEMM 2 k^2 Cin Cout OH OW, Winograd
    F(2x2,3x3) 2 * 16 Cin Cout per 2x2 output tile (2.25x fewer) — plus the fused correlation's 13.92 GFLOP (CORR_FLOPS)."""
    from dvc_amd import ops
    ops.conv_record = rec = []
    try:
        cc.frame(frame, last, graph=False)
        torch.cuda.synchronize()
    finally:
        ops.conv_record = None
    executed = direct = 0.0
    for r in rec:
        OH, OW = ops.conv_out_hw(r["H"], r["W"], r["ksize"], r["stride"], r["dil"], r["pad"], r["in_up"], r["in_sub"])
        d = 2.0 * r["ksize"] ** 2 * r["Cin"] * r["Cout"] * OH * OW * r["N"]
        direct += d
        if str(r.get("algo", "")).startswith("winograd"):
            executed += 2.0 * 16 * r["Cin"] * r["Cout"] * ((OH + 1) // 2) * ((OW + 1) // 2) * r["N"]
        else:
            executed += d
    return executed + CORR_FLOPS, direct + CORR_FLOPS, len(rec)


def torchrun_command(n, argv, port, environ=None):
    """(argv, environment) of the N-rank launch `python bench.py --gpus N` performs on itself: one process per GPU under
    torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve), the caller's own
    flags passed through unchanged, dmabuf IPC for RCCL (the host driver supports no other)."""
    environ = os.environ if environ is None else environ
    env = dict(environ, HSA_ENABLE_IPC_MODE_LEGACY=environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return cmd, env


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start N ranks ourselves."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd, env = torchrun_command(n, sys.argv[1:], port)
    log("[bench] launching", " ".join(cmd))
    return subprocess.call(cmd, env=env)


def other_config_leg(nets, device, h, w, corr, K, reps, lookahead, use_graph):
    """One of BASELINE.json's other single-GPU configurations, timed inside this run (r05 review: configs[3] and configs[4] only
    had builder-run lines): a K-frame clip at h x w with the `corr` correlation through the same pipelined driver as `value`,
    `reps` times, median; plus the configuration's correlation launch set on the clip's own operands.  Warm-up (autotune of
    the new shapes, graph capture, stream pools) is untimed.  Short by design — the headline's repeats / p10 / p90 machinery
    stays with the headline; the stand-alone lines under profiles/ (bench.py --hw 432x768 / --corr bf16) are the long form."""
    from dvc_amd import ops, synth
    from dvc_amd.frame import VGG_OUT, ClipColorizer
    from dvc_amd.util import feature_normalize, gray2rgb_batch
    vgg, warp, _ = nets
    keep = warp.corr_precision
    warp.corr_precision = corr
    try:
        cc = ClipColorizer(*nets, temperature=1e-10, graph=use_graph)
        cc.set_exemplar(synth.synth_lab(synth.EXEMPLAR_SEED, h, w).to(device))
        Wm = 3
        frames = [synth.synth_lab(synth.FRAME_SEED0 + i, h, w).to(device) for i in range(K + Wm)]
        for _ in range(2):
            cc.clip(frames[:Wm], lookahead=lookahead)
        last = cc.last_lab

        def run():
            cc.clip(frames[Wm:], last=last, lookahead=lookahead)
            return cc.last_lab
        run()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = run()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        assert torch.isfinite(out).all(), "non-finite output"
        t = sorted(ts)[len(ts) // 2]
        # the configuration's correlation stage on the clip's own operands (HIP events on the launch stream)
        p = (h // 4) * (w // 4)
        flops = 2.0 * p * p * C + 2.0 * p * p * 3
        fA = vgg(gray2rgb_batch(frames[Wm][:, 0:1]), VGG_OUT)
        fe = warp.features(*[feature_normalize(x) for x in fA[1:]])
        bf16 = corr == "bf16"
        th = warp.project("theta", fe, bf16=bf16)
        ph, bl4 = cc.ex_cache
        bl = bl4.view(1, 3, -1)
        if bf16:
            launch = lambda: ops.corr_fwd_bf16(th, ph, bl, 1e-10, h // 4, w // 4)       # noqa: E731
        else:
            launch = lambda: ops.corr_fwd(th, ph, bl, 1e-10, h // 4, w // 4, defer_merge=ops.fold_merge())   # noqa: E731
        n_warm, n_rep = (100, 50) if p <= 6000 else (10, 10)
        for _ in range(n_warm):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_rep):
            launch()
        e1.record()
        torch.cuda.synchronize()
        t_corr = e0.elapsed_time(e1) * 1e-3 / n_rep
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        return {"frames_per_s": round(K / t, 3), "ms_per_step": round(t / K * 1e3, 4), "steps": K, "repeats": reps,
                "ms_per_step_min_max": [round(min(ts) / K * 1e3, 4), round(max(ts) / K * 1e3, 4)],
                "corr_launch_us": round(t_corr * 1e6, 2), "corr_frac": round(flops / t_corr / 1e12 / peak, 4),
                "corr_peak_tflops": peak, "corr_gflop": round(flops / 1e9, 2)}
    finally:
        warp.corr_precision = keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-speed-leg", action="store_true", help="skip the config.engine_speed leg (the K frames under DVC_CONV_ALGO=speed)")
    ap.add_argument("--no-parity", action="store_true", help="skip the untimed parity block (GPU and CPU fp32 against the fp64 oracle)")
    ap.add_argument("--clock-warmup-s", type=float, default=0.5,
                    help="seconds of untimed load (the warm-up frames, repeated) before the W warm-up steps, so that the "
                         "timed region does not start on a ramping GPU clock; 0 disables")
    ap.add_argument("--hw", default="216x384",
                    help="frame size HxW; the default is BASELINE configs[1] (the metric's configuration), "
                         "432x768 is configs[3] (information only: no CPU baseline, traffic not re-measured)")
    ap.add_argument("--front-batch", type=int, default=1,
                    help="frames per front-end batch of the clip driver (bit-identical to 1: the library plans per image)")
    ap.add_argument("--lookahead", type=int, default=2,
                    help="frames whose front end runs ahead on side HIP streams (0 = per-frame calls on one stream)")
    ap.add_argument("--no-graph", action="store_true",
                    help="issue every kernel launch from Python instead of replaying the captured per-frame launch sequences "
                         "(hipGraph, dvc_amd/graph.py); results are bit-identical either way")
    ap.add_argument("--autotune", action="store_true",
                    help="time the general direct engine's tile / split candidates on first use (the analogue of cudnn.benchmark = True, "
                         "test.py:140) instead of the library's static plan.  Off by default since r06: the tuner's picks differ "
                         "from run to run, every pick is another summation order, and with the chaotic random weights that "
                         "re-draws the tail of a frame's error field (profiles/r06_parity_pool_probe.txt) — with the static plan "
                         "the timed arithmetic is EXACTLY the arithmetic the parity tests assert; it is worth +0.4 % frames/s")
    ap.add_argument("--no-autotune", action="store_true", help="(default since r06; kept for old command lines)")
    ap.add_argument("--min-timed-s", type=float, default=0.5,
                    help="the K-step clip is repeated until at least this much GPU work has been timed (>= 3 repeats); the line "
                         "reports the median repeat with p10 / p90")
    ap.add_argument("--clips", type=int, default=4,
                    help="B of the batched-clips leg (config.batched_clips: frame t of B independent clips, each with its own "
                         "exemplar, per step — the serving form of the path; batch-aware launch plan); 0 skips the leg")
    ap.add_argument("--refs", type=int, default=4,
                    help="R of the multi-reference leg (config.multi_reference: the same clip against R exemplars in one pass, "
                         "test.py:169-181); 0 skips the leg")
    ap.add_argument("--corr", choices=["fp32", "bf16"], default="fp32",
                    help="bf16 = BASELINE configs[4]: bf16 MFMA candidate filter + exact fp32 re-scoring")
    ap.add_argument("--other-steps", type=int, default=20,
                    help="steps of the config.other_configs legs (BASELINE configs[3] 432x768 and configs[4] bf16 correlation, timed "
                         "in the default single-GPU 216x384 fp32 run, 3 repeats each); 0 skips them")
    ap.add_argument("--no-exemplar-cache", action="store_true",
                    help="recompute the exemplar side of WarpNet every frame, as the reference does")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The host side of the GPU legs (synthesising the frames, issuing ~150 launches per frame from one thread) must not be
    # starved by its own helper threads: the GPU boxes expose 256 hardware threads under a 16-CPU cgroup quota, and an ATen
    # pool sized for the mask burns the quota — the launch thread then gets throttled in the middle of the timed region
    # (seen as a 380 vs 415 frames/s difference between otherwise identical runs).  cpu_baseline() sets its own counts.
    torch.set_num_threads(max(1, min(16, host_cpus()[0] // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # launched by torch.distributed.run
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.gpus < 1:
        sys.exit("[bench] --gpus must be >= 1")
    if torch.cuda.device_count() < args.gpus:
        sys.exit(f"[bench] --gpus {args.gpus} requested but only {torch.cuda.device_count()} GPU(s) visible: refusing to "
                 "print a line for a different n_gpus")
    if not use_dist and args.gpus > 1:
        sys.exit(relaunch_under_torchrun(args.gpus))
    if use_dist and world != args.gpus:
        sys.exit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for a different n_gpus")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    global H, W, P, CORR_FLOPS, CORR_BYTES, PATH_FLOPS
    if args.hw != "216x384":
        H, W = (int(v) for v in args.hw.lower().split("x"))
        assert H % 16 == 0 and W % 16 == 0, "--hw: multiples of 16"
        scale = (H * W) / (216.0 * 384.0)
        P = (H // 4) * (W // 4)
        CORR_FLOPS = 2.0 * P * P * C + 2.0 * P * P * 3
        CORR_BYTES = 4.0 * (2 * P * C + 3 * P + 3 * P + P)
        PATH_FLOPS = (348.4e9 - 13.92e9) * scale + CORR_FLOPS     # convolutions scale with the pixels
        args.no_cpu_baseline = True
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if use_dist:
        # RCCL prints a version banner on STDOUT at communicator creation; keep stdout for the ONE JSON
        # line by pointing fd 1 at stderr while the communicator is created (first collective).
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    n_gpus = world

    from dvc_amd import ops, synth
    from dvc_amd.frame import ClipColorizer
    from dvc_amd.parallel import broadcast_exemplar

    ops.set_autotune(bool(args.autotune) and not args.no_autotune)
    nets, sd = build_nets(device)
    nets[1].corr_precision = args.corr
    use_graph = not args.no_graph and not args.no_exemplar_cache and args.front_batch == 1
    cc = ClipColorizer(*nets, temperature=1e-10, cache_exemplar=not args.no_exemplar_cache, graph=use_graph)
    graph_note = None
    # exemplar: prepared on rank 0, shared once with every rank (RCCL broadcast over xGMI)
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(device)
    broadcast_exemplar(cc, IB if rank == 0 else None, (1, 3, H, W), device, src=0)

    if use_graph:
        # stream capture is exercised here, before anything is timed: if the runtime refuses it on this box the line is still
        # produced — with every launch issued from Python, and saying so — instead of no line at all (results are identical)
        try:
            probe = synth.synth_lab(synth.FRAME_SEED0, H, W).to(device)
            cc.frame(probe, torch.zeros_like(probe))
            cc.clip([probe, probe, probe], lookahead=max(args.lookahead, 1))
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            graph_note = f"hipGraph capture failed ({type(e).__name__}: {e}); fell back to launches from Python"
            log("[bench]", graph_note)
            use_graph = False
            cc.graph = False            # (same driver, same exemplar state: no collective on this path)
            cc._graphs.clear()

    K, Wm = args.steps, args.warmup
    # this rank's contiguous chunk of the clip: warm-up frames then K timed frames (resident in HBM)
    base = synth.FRAME_SEED0 + rank * (K + Wm)
    frames = [synth.synth_lab(base + i, H, W).to(device) for i in range(K + Wm)]
    last = torch.zeros(1, 3, H, W, device=device)

    def step(i, last):
        ab, _ = cc.frame(frames[i], last)
        return torch.cat((frames[i][:, 0:1], ab), dim=1)      # test.py:96

    # clock warm-up (untimed, before the W warm-up steps): a GPU that has just been handed to this process takes a few
    # hundred milliseconds of load to reach its sustained clocks, and W x 2.5 ms is far less — without this the first
    # frames of a short timed region run on a ramping clock (measured: 370 frames/s at --steps 30 against 411 at --steps 60
    # in otherwise identical runs).  Same frames, same kernels as the warm-up steps; results discarded.
    if Wm > 0:
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < args.clock_warmup_s:
            lw = torch.zeros(1, 3, H, W, device=device)
            for i in range(Wm):
                lw = step(i, lw)
            torch.cuda.synchronize()
    for i in range(Wm):
        last = step(i, last)
    # launch mode of the timed region: FIXED (r04) — look-ahead front ends replayed as hipGraphs, ColorVidNet chain launched
    # kernel by kernel (ClipColorizer.graph_parts); the other mode (every launch from Python) is timed after the timed region
    # and reported next to it.  (r03 chose the faster of the two on a warm-up trial: a best-of selection.)
    clip_graph = use_graph
    if args.lookahead > 0:
        # second untimed pass over the warm-up frames through the clip driver: the side streams' memory
        # pools (and nothing else) are still cold after the per-frame pass above
        cc.clip(frames[:Wm], lookahead=args.lookahead, front_batch=args.front_batch, graph=clip_graph)

    def timed_clip():
        """Exactly K steps of the recurrence from the state after the warm-up frames; returns the final [L, ab]."""
        if args.lookahead > 0:
            # the clip driver: front end (VGG19 + WarpNet + correlation) of frames t+1.. on side HIP streams while
            # this stream runs the ColorVidNet recurrence; bit-identical to the per-frame loop below
            cc.clip(frames[Wm:Wm + K], last=last, lookahead=args.lookahead, front_batch=args.front_batch, graph=clip_graph)
            return cc.last_lab
        lt = last
        for i in range(Wm, Wm + K):
            lt = step(i, lt)
        return lt

    def bracket():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # one untimed pass of the K steps: sizes the number of repeats (the same on every rank) and is the last piece of warm-up
    bracket()
    t0 = time.perf_counter()
    timed_clip()
    bracket()
    est = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(est, op=dist.ReduceOp.MAX)
    repeats = int(min(400, max(3, -(-args.min_timed_s // max(est.item(), 1e-6)))))
    rep_s = []
    for _ in range(repeats):
        bracket()
        t0 = time.perf_counter()
        last_timed = timed_clip()
        bracket()
        rep_s.append(time.perf_counter() - t0)
    def median_s(fn, n):
        """Median wall time of `fn()` over n runs (each synchronised), and its last result."""
        ts, res = [], None
        for _ in range(n):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            res = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        return sorted(ts)[len(ts) // 2], res

    side_reps = max(1, min(repeats, 5))
    # the same K frames through the reference's per-frame API (frame_colorization called frame by frame, no
    # look-ahead possible): reported next to `value`, and checked to give the same predictions
    seq_fps = None
    if args.lookahead > 0 and rank == 0:
        def per_frame_loop():
            lt = last
            for i in range(Wm, Wm + K):
                lt = step(i, lt)
            return lt
        t_seq, last_seq = median_s(per_frame_loop, side_reps)
        seq_fps = K / t_seq
        assert torch.equal(last_seq, last_timed), "pipelined clip driver != per-frame loop"
    # ... and through the reference's UNMODIFIED call pattern (test.py:57-96): `features_B` computed once by the caller, then
    # `frame_colorization(IA_lab, IB_lab, I_last, features_B, vggnet, nonlocal_net, colornet, feature_noise=0, temperature=1e-10)`
    # frame by frame with the same exemplar tensors — no ClipColorizer, no cache argument, every launch from Python.  The
    # exemplar side is memoised behind the call (nets.WarpNet._memo_exemplar_side); with DVC_EXEMPLAR_MEMO=0 it is recomputed per
    # frame as the reference does (timed next to it).
    dropin = None
    if args.lookahead > 0 and rank == 0 and n_gpus == 1 and args.corr == "fp32":     # (single-GPU legs: the N > 1 runs time `value` only)
        from models.FrameColor import frame_colorization
        from utils.util import tensor_lab2rgb, uncenter_l
        vggnet, nonlocal_net, colornet = nets
        I_reference_lab = IB
        I_reference_rgb = tensor_lab2rgb(torch.cat((uncenter_l(I_reference_lab[:, 0:1]), I_reference_lab[:, 1:3]), dim=1))
        features_B = vggnet(I_reference_rgb, ["r12", "r22", "r32", "r42", "r52"], preprocess=True)

        def reference_loop():
            I_last_lab_predict = last
            for i in range(Wm, Wm + K):
                IA_lab = frames[i]
                IA_l = IA_lab[:, 0:1, :, :]
                I_current_ab_predict, _, _ = frame_colorization(IA_lab, I_reference_lab, I_last_lab_predict, features_B, vggnet,
                                                                nonlocal_net, colornet, feature_noise=0, temperature=1e-10)
                I_last_lab_predict = torch.cat((IA_l, I_current_ab_predict), dim=1)
            return I_last_lab_predict
        reference_loop()
        t_drop, last_drop = median_s(reference_loop, side_reps)
        drop_diff = (last_drop - last_timed).abs().max().item()
        assert drop_diff <= 1e-4, f"unmodified reference loop != clip driver ({drop_diff})"
        ops.set_exemplar_memo(False)
        try:
            reference_loop()
            t_drop_nomemo, _ = median_s(reference_loop, max(1, min(side_reps, 3)))
        finally:
            ops.set_exemplar_memo(True)
        dropin = {"frames_per_s": round(K / t_drop, 3), "exemplar_side_recomputed_per_frame_frames_per_s": round(K / t_drop_nomemo, 3),
                  "bit_identical_to_clip_driver": bool(drop_diff == 0.0)}
    # ... and, when the timed region replayed captured launch sequences, the same K frames with every launch issued from
    # Python (what r01/r02 timed): reported next to `value`, and required to give the same predictions bit for bit
    eager_fps = None
    if use_graph and args.lookahead > 0 and rank == 0:
        cc.clip(frames[:Wm], lookahead=args.lookahead, graph=False)          # (side-stream allocator pools)

        def eager_clip():
            cc.clip(frames[Wm:Wm + K], last=last, lookahead=args.lookahead, graph=False)
            return cc.last_lab
        t_eager, last_eager = median_s(eager_clip, side_reps)
        eager_fps = K / t_eager
        assert torch.equal(last_eager, last_timed), "hipGraph replay != eager launches"
    # ... and the multi-reference pass (test.py:169-181 colourises the same clip once per reference image): the same K frames
    # against R exemplars at once — one front end per frame, R correlations, ColorVidNet chain at batch R (set_exemplars)
    multi = None
    if args.refs > 1 and args.lookahead > 0 and rank == 0 and not args.no_exemplar_cache:
        ref_seeds = [synth.EXEMPLAR_SEED, 3, 5, 11, 13, 17, 19, 23][:args.refs]
        # (every launch from Python: with four correlations per front end a replayed front-end graph is 2 % SLOWER here —
        # 626 vs 634-641 frame-colourisations/s, profiles/r04_refs_chain_probe.txt — its backlog of packets delays the chain)
        cc_m = ClipColorizer(*nets, temperature=1e-10)
        cc_m.set_exemplars([synth.synth_lab(sd_, H, W).to(device) for sd_ in ref_seeds])
        cc_m.clip(frames[:max(Wm, 2)], lookahead=args.lookahead)             # warm-up (autotune at batch R, stream pools)

        def refs_clip():
            return cc_m.clip(frames[Wm:Wm + K], lookahead=args.lookahead)[-1]
        t_m, ab_m = median_s(refs_clip, side_reps)
        assert torch.isfinite(ab_m).all()
        multi = {"R": len(ref_seeds), "frame_colorizations_per_s": round(len(ref_seeds) * K / t_m, 3),
                 "clip_frames_per_s": round(K / t_m, 3), "ms_per_clip_frame": round(t_m / K * 1e3, 4),
                 "note": "the K timed frames against R exemplars in one pass (ClipColorizer.set_exemplars): one VGG19 + WarpNet "
                         "front end per frame, R fused correlations, ColorVidNet at batch R with a batch-aware launch plan; "
                         "compare frame_colorizations_per_s with `value` (R passes, one reference each)"}
    # ... and B independent clips in lock step (serving): every step colourises frame t of B clips, each against its own exemplar;
    # front ends and chain at batch B, planned for the batch (ClipColorizer(batch_plan=True))
    batched = None
    if args.clips > 1 and args.lookahead > 0 and rank == 0 and not args.no_exemplar_cache and (H, W) == (216, 384):
        Bc = args.clips
        cc_b = ClipColorizer(*nets, temperature=1e-10, batch_plan=True)
        cc_b.set_exemplar(torch.cat([synth.synth_lab(synth.EXEMPLAR_SEED + 10 * c, H, W) for c in range(Bc)]).to(device))
        bframes = [torch.cat([synth.synth_lab(synth.FRAME_SEED0 + 1000 * c + i, H, W) for c in range(Bc)]).to(device)
                   for i in range(K + max(Wm, 2))]
        cc_b.clip(bframes[:max(Wm, 2)], lookahead=args.lookahead)

        def batched_clip():
            return cc_b.clip(bframes[max(Wm, 2):], lookahead=args.lookahead)[-1]
        t_b, ab_b = median_s(batched_clip, side_reps)
        assert torch.isfinite(ab_b).all() and tuple(ab_b.shape) == (Bc, 2, H, W)
        batched = {"B": Bc, "frames_per_s": round(Bc * K / t_b, 3), "ms_per_step": round(t_b / K * 1e3, 4),
                   "note": "frame t of B independent clips per step, each clip with its own exemplar (the batched form of "
                           "frame_colorization, train.py:402 calls it with B = 16): front ends and ColorVidNet chain at batch B with the "
                           "batch-aware launch plan; per-clip results equal the single-clip driver's to fp32 rounding of the summation "
                           "order (tests/test_gpu_refs.py)"}
    last_before = last          # recurrence state the timed clip started from
    last = last_timed
    t = torch.tensor(rep_s, device=device, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)        # per repeat: the slowest rank
    rep_sorted = sorted(t.tolist())
    pick = lambda q: rep_sorted[min(len(rep_sorted) - 1, max(0, int(round(q * (len(rep_sorted) - 1)))))]   # noqa: E731
    elapsed = pick(0.5)                                 # the MEDIAN repeat of exactly K steps
    assert torch.isfinite(last).all(), "non-finite output"
    fps = n_gpus * K / elapsed

    # ---- roofline of the north-star kernel: HIP events on the launch stream, on the CLIP'S OWN operands — theta of
    # the first timed frame (VGG19 -> WarpNet heads/trunk -> 1x1 -> centre/normalise, exactly what frame_colorization
    # feeds the kernel), phi and the pooled Lab of the exemplar
    roof = None
    if rank == 0:
        from dvc_amd.frame import VGG_OUT
        from dvc_amd.util import feature_normalize, gray2rgb_batch
        vgg, warp, _ = nets
        fr = frames[Wm]
        fA = vgg(gray2rgb_batch(fr[:, 0:1]), VGG_OUT)
        th = warp.project("theta", warp.features(*[feature_normalize(t) for t in fA[1:]]))
        if cc.ex_cache is not None and not isinstance(cc.ex_cache[0], tuple):
            ph, bl4 = cc.ex_cache
        else:   # (--no-exemplar-cache / bf16 cache layout: rebuild the fp32 exemplar side for this leg)
            ph, bl4 = warp.exemplar_side(cc.IB_lab, *[feature_normalize(t) for t in cc.features_B[1:]], bf16=False)
        bl = bl4.view(1, 3, -1)
        # warm-up: the first milliseconds after the clip run at a lower clock (measured: 148 us vs 129 us per launch once the
        # correlation alone has run for ~10 ms — tools/corr_ab_probe.py times round-robin for the same reason)
        # What is timed is what the clip driver launches for this stage: since r04 the fused kernel alone — the merge of its
        # per-workgroup partial states runs inside the consumer's launch (ops.pack_color_input -> dvc_corr_merge_pack, which
        # replaces the separate merge AND pack launches) — unless DVC_FOLD_MERGE=0 / the bf16 path; the stand-alone form
        # (kernel + corr_merge_kernel, what r01-r03 timed) is measured right after and reported next to it.
        folded = ops.fold_merge() and args.corr == "fp32"
        reps = 50 if P <= 6000 else 10

        def time_corr(defer):
            for _ in range(100 if P <= 6000 else 10):
                ops.corr_fwd(th, ph, bl, 1e-10, H // 4, W // 4, defer_merge=defer)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                      # ops launch on torch's current stream, so these events see them
            for _ in range(reps):
                ops.corr_fwd(th, ph, bl, 1e-10, H // 4, W // 4, defer_merge=defer)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / reps
        t_standalone = time_corr(False)      # fused kernel + its merge kernel
        t_corr = time_corr(True) if folded else t_standalone
        achieved = CORR_FLOPS / t_corr / 1e12
        bf16_roof = None
        if args.corr == "bf16":
            # configs[4]: the correlation this run's frames went through is the bf16 candidate filter + exact fp32 re-scoring
            # (csrc/corr_bf16.hip: two bf16 MFMA sweeps, then one wave per query re-scores its candidates in fp32), not
            # corr_fwd_kernel — time THAT set of launches on the clip's own operands, against the bf16 matrix peak.  It is an
            # exactness-preserving filter, not a bf16-rate kernel: no throughput claim rides on it.
            th16 = warp.project("theta", warp.features(*[feature_normalize(t) for t in fA[1:]]), bf16=True)
            ph16, bl16 = cc.ex_cache if isinstance(cc.ex_cache[0], tuple) else warp.exemplar_side(
                cc.IB_lab, *[feature_normalize(t) for t in cc.features_B[1:]], bf16=True)

            def time_bf16():
                for _ in range(100 if P <= 6000 else 10):
                    ops.corr_fwd_bf16(th16, ph16, bl16.view(1, 3, -1), 1e-10, H // 4, W // 4)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.corr_fwd_bf16(th16, ph16, bl16.view(1, 3, -1), 1e-10, H // 4, W // 4)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e-3 / reps
            t16 = time_bf16()
            bf16_roof = {"kernel": "corr_bf16 pass kernels (two bf16 MFMA sweeps) + fp32 re-scoring kernel + merge, all launches of "
                                   "one dvc_corr_fwd_bf16 call", "bound": "mfma", "achieved": round(CORR_FLOPS / t16 / 1e12, 3),
                         "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(CORR_FLOPS / t16 / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                         "avg_launch_us": round(t16 * 1e6, 2), "traffic": None,
                         "note": "algorithmic 13.92 GFLOP of the stage / time of the whole launch set; the per-kernel split (sweeps vs "
                                 "re-scoring) is in the committed rocprofv3 trace of this command, profiles/r05_bench_bf16_kernel_stats.csv; "
                                 "an exactness-preserving candidate filter (results identical to the fp32 kernel's), not a bf16-rate "
                                 "kernel: no throughput claim (DESIGN.md 4.1b)",
                         "fp32_kernel_for_comparison": {"kernel": "corr_fwd_kernel", "avg_launch_us": round(t_corr * 1e6, 2),
                                                        "frac_of_fp32_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4)}}
        traffic, traffic_src = corr_traffic() if (H, W) == (216, 384) else (None, {"kind": "not measured at this size"})
        exec_flops, direct_flops, n_convs = executed_matrix_flops(cc, frames[Wm], torch.zeros_like(frames[Wm]))
        exec_tflops = exec_flops * fps / n_gpus / 1e12
        roof = {"kernel": "corr_fwd_kernel (merge folded into the consumer's launch, dvc_corr_merge_pack)" if folded else
                "corr_fwd_kernel (+corr_merge_kernel)", "bound": "mfma",
                "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "operands": "theta of the first timed frame, phi / pooled Lab of the exemplar (the clip's own features)",
                "avg_launch_us": round(t_corr * 1e6, 2),
                "with_standalone_merge": {"avg_launch_us": round(t_standalone * 1e6, 2),
                                          "frac": round(CORR_FLOPS / t_standalone / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                          "note": "corr_fwd_kernel + corr_merge_kernel as two launches: the form r01-r03 timed"},
                "hbm_view": {"achieved": round(CORR_BYTES / t_corr / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": round(CORR_BYTES / t_corr / 1e9 / PEAK_HBM_GBS, 5),
                             "note": "fused kernel never materialises the PxP affinity; compulsory bytes "
                                     "10.76 MB/frame make it MFMA-bound, not HBM-bound (SURVEY.md 8d)"},
                "whole_path": {"executed": {"achieved": round(exec_tflops, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                            "frac": round(exec_tflops / PEAK_F32_MFMA_TFLOPS, 4),
                                            "gflop_per_frame": round(exec_flops / 1e9, 2),
                                            "flop_count": f"matrix FLOPs the kernels EXECUTE per frame ({n_convs} convolution launches "
                                                          "recorded from a per-frame call and priced by the engine that ran them: "
                                                          "Winograd F(2x2,3x3) 16 instead of 36 multiplications per 2x2 tile and channel "
                                                          "pair; + 13.92 GFLOP correlation) x frames/s / fp32 MFMA peak"},
                               "effective_tflops": round(PATH_FLOPS * fps / n_gpus / 1e12, 3),
                               "effective_note": "348.4 GFLOP per 216x384 frame (SURVEY.md 8d: algorithmic FLOPs with DIRECT "
                                                 "convolutions; counted here: %.1f) x frames/s - NOT a roofline fraction: the "
                                                 "Winograd layers execute 2.25x fewer multiplications" % (direct_flops / 1e9)}}

    if rank == 0 and roof is not None and bf16_roof is not None:
        roof = dict(bf16_roof, hbm_view=roof["hbm_view"], whole_path=roof["whole_path"])
    # ---- the same K frames under the geometry-only Winograd rule ("speed": what r01-r04 timed as the default): reported next to
    # `value`, never as `value` — at this configuration that engine is further from the fp64 truth than the reference's own CPU
    # fp32 run (parity.engine_speed_for_comparison), which is why the default keeps arch.DIRECT_LAYERS on the direct engine
    speed_leg = None
    if rank == 0 and n_gpus == 1 and args.lookahead > 0 and ops.conv_algo() == "auto" and ops.direct_layers() and not args.no_speed_leg:
        try:
            ops.set_conv_algo("speed")
            cc.clip(frames[:max(Wm, 3)], lookahead=args.lookahead, graph=clip_graph)      # re-capture / autotune under this engine
            cc.clip(frames[:max(Wm, 3)], lookahead=args.lookahead, graph=clip_graph)

            def speed_clip():
                cc.clip(frames[Wm:Wm + K], last=last_before, lookahead=args.lookahead, graph=clip_graph)
                return cc.last_lab
            t_sp, last_sp = median_s(speed_clip, side_reps)
            assert torch.isfinite(last_sp).all()
            speed_leg = {"frames_per_s": round(K / t_sp, 3), "ms_per_step": round(t_sp / K * 1e3, 4),
                         "note": "Winograd F(2x2,3x3) on every eligible layer (geometry rule only, DVC_CONV_ALGO=speed): the engine "
                                 "choice of r01-r04, timed here with the same driver on the same frames; NOT the headline — see "
                                 "parity.engine_speed_for_comparison for what it costs in accuracy"}
        finally:
            ops.set_conv_algo("auto")
            cc.clip(frames[:max(Wm, 3)], lookahead=args.lookahead, graph=clip_graph)      # back to the default engine's sequences
    cpu = parity = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        # (the two host-CPU legs must not take the measured line with them: a failure is reported in their place)
        try:
            cpu = cpu_baseline(sd)
        except Exception as e:      # noqa: BLE001
            cpu = {"value": None, "unit": "frames/s", "cores": None, "kind": "port", "error": f"{type(e).__name__}: {e}"}
            log("[bench] cpu baseline failed:", cpu["error"])
        if (H, W) == (216, 384) and not args.no_parity:
            try:
                parity = parity_block(cc, sd, device)
                log(f"[bench] parity vs fp64, pooled over {parity['frames_pooled']} frames: GPU / CPU fp32 {parity['gpu_over_cpu32']}; frame 1000 "
                    f"{parity['frame_1000']['gpu_over_cpu32']}; the reference at another thread count "
                    f"{parity['cpu32_other_thread_count']['over_cpu32']}")
            except Exception as e:      # noqa: BLE001
                parity = {"error": f"{type(e).__name__}: {e}"}
                log("[bench] parity block failed:", parity["error"])

    other = None
    if (rank == 0 and n_gpus == 1 and (H, W) == (216, 384) and args.corr == "fp32" and args.lookahead > 0 and args.other_steps > 0
            and not args.no_exemplar_cache):
        t_o = time.perf_counter()
        other = {}
        for name, (h_, w_, corr_) in (("432x768", (432, 768, "fp32")), ("bf16", (216, 384, "bf16"))):
            try:
                other[name] = other_config_leg(nets, device, h_, w_, corr_, args.other_steps, 3, args.lookahead, use_graph)
            except Exception as e:      # noqa: BLE001  (a leg that fails must not take the headline line with it)
                other[name] = {"error": f"{type(e).__name__}: {e}"}
            log(f"[bench] other config {name}: {other[name]}")
        other["432x768"]["workload"] = ("BASELINE configs[3]: 1x3x432x768 frame + one exemplar per step (P = 20736 correlation "
                                        "positions), same driver and engine choice as `value`, fp32")
        other["bf16"]["workload"] = ("BASELINE configs[4], 1 GPU: 216x384, correlation = bf16 MFMA candidate filter + exact fp32 "
                                     "re-scoring (results identical to the fp32 kernel's at T <= 1e-4; an exactness-preserving "
                                     "filter, no bf16-rate claim: corr_frac is against the dense bf16 peak)")
        other["note"] = (f"{args.other_steps} steps x 3 repeats each (median), warm-up with autotune / capture untimed; "
                         f"{time.perf_counter() - t_o:.1f} s of this run; corr_launch_us / corr_frac: the configuration's correlation "
                         "launch set on the clip's own operands, HIP events")

    if rank == 0:
        line = {
            "metric": f"colorized frames/sec/GPU at {H}x{W}; correlation HBM GB/s vs roofline",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": n_gpus, "steps": K, "warmup": Wm,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: 1x3x216x384" if (H, W) == (216, 384) else f"1x3x{H}x{W}") +
                                   " frame + one exemplar per step, HIP "
                                   "VGG19 + WarpNet/fused correlation + ColorVidNet forward, fp32, clip "
                                   "recurrence as test.py:68-96",
                       "H": H, "W": W, "temperature": 1e-10, "weights": "synthetic seed 0",
                       "correlation": "fp32 MFMA" if args.corr == "fp32" else
                       "bf16 MFMA candidate filter + exact fp32 re-scoring (configs[4])",
                       "exemplar_side": "recomputed per frame" if args.no_exemplar_cache else "cached per clip",
                       "conv_algorithm": {"auto": "Winograd F(2x2,3x3) on the fp32 matrix cores where ops.winograd_selected "
                                                  "picks it (3x3 stride-1 layers with >= 13x24 outputs, minus the layers of the "
                                                  "error-aware engine map arch.DIRECT_LAYERS: %s), direct implicit GEMM elsewhere; "
                                                  "fp32 throughout" % (",".join(sorted(ops.direct_layers())) or "none"),
                                          "speed": "Winograd F(2x2,3x3) on every 3x3 stride-1 layer with >= 13x24 outputs (geometry "
                                                   "rule only, no error-aware map)",
                                          "winograd": "Winograd F(2x2,3x3) on every eligible 3x3 layer",
                                          "direct": "direct implicit GEMM everywhere"}[ops.conv_algo()],
                       "conv_tile_choice": "autotuned on first use during warm-up (cf. cudnn.benchmark=True, test.py:140)"
                       if ops.autotune_enabled() else "the library's static plan (deterministic: the arithmetic the parity tests assert)",
                       "frames_per_gpu": K, "parallelism": f"frame-chunks x{n_gpus}",
                       "repeats": repeats, "ms_per_step_p10": round(pick(0.1) / K * 1e3, 4),
                       "ms_per_step_p90": round(pick(0.9) / K * 1e3, 4),
                       "timed_gpu_seconds": round(sum(rep_sorted), 3),
                       "timing": f"the K = {K} step clip timed {repeats} times back to back from the same recurrence state, every "
                                 "repeat bracketed by barrier + synchronize; ms_per_step / value = the median repeat",
                       "clip_driver": "per-frame calls, one stream" if args.lookahead <= 0 else
                       f"ClipColorizer.clip: front end of the next {args.lookahead} frames on side HIP streams, " +
                       (f"{args.front_batch} frames per set of front-end launches (planned per image), " if args.front_batch > 1 else "") +
                       "ColorVidNet recurrence on the main stream (bit-identical to per-frame calls)",
                       "launches": ("look-ahead front ends replayed as hipGraphs (one captured sequence per side stream), ColorVidNet "
                                    "chain launched kernel by kernel; per-frame API: both sequences replayed; bit-identical to "
                                    "eager launches (fixed mode, no warm-up trial)") if use_graph else
                                   (graph_note or "every kernel launched from Python"),
                       "engine_speed": speed_leg,
                       "per_frame_api_frames_per_s": None if seq_fps is None else round(seq_fps, 3),
                       "dropin_unmodified_frames_per_s": None if dropin is None else dropin["frames_per_s"],
                       "dropin_unmodified": None if dropin is None else dict(
                           dropin, note="the reference's own loop, test.py:57-96, verbatim: features_B computed by the caller, "
                                        "frame_colorization(...) called positionally per frame with the same exemplar tensors, no "
                                        "ClipColorizer in the caller, every launch from Python; the exemplar side is memoised behind "
                                        "the call on the identity / version counters of the caller's tensors (bit-identical to "
                                        "recomputing it; tests/test_gpu_dropin_loop.py)"),
                       "eager_launch_clip_driver_frames_per_s": None if eager_fps is None else round(eager_fps, 3),
                       "batched_clips": batched,
                       "other_configs": other,
                       "multi_reference": None if multi is None else dict(
                           multi, speedup_vs_one_pass_per_reference=round(multi["frame_colorizations_per_s"] / (fps / n_gpus), 3))},
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
