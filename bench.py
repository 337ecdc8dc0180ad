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
N=1 runs BASELINE.json configs[1].  With N>1 each rank colourises its own contiguous chunk of K frames (weak
scaling; the exemplar-side tensors are computed on rank 0 and broadcast once over RCCL/xGMI; no collective in
the per-frame path).  Rank 0 prints ONE JSON line on stdout; diagnostics go to stderr.
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
PEAK_HBM_GBS = 8000.0
# PMC-measured HBM traffic of one corr_fwd_kernel launch at P=5184 (profiles/r01_pmc_summary.md):
# 102.2 MB read + 3.7 MB written vs 10.76 MB compulsory (phi is re-streamed through the per-XCD L2s)
CORR_TRAFFIC_BYTES = 105.9e6
CORR_TRAFFIC_SOURCE = "profiles/r01_pmc_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"


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


def cpu_baseline(sd, n_timed=4):
    """Oracle (torch-CPU restatement of the reference, bit-exact vs the reference modules — see
    oracle/pin_reference.py) timed on this box's host cores on the same workload."""
    from dvc_amd import synth
    from oracle import dvc_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # ATen's CPU kernels stop scaling (and oversubscribe badly) far below a 256-thread host;
    # use at most 32 threads and say so.
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    torch.set_flush_denormal(True)
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    with torch.no_grad():
        fB = O.exemplar_features(IB, sd[0])
        last = torch.zeros(1, 3, H, W)
        times = []
        for i in range(1 + n_timed):
            fr = synth.synth_lab(synth.FRAME_SEED0 + i, H, W)
            t0 = time.perf_counter()
            ab, _, _ = O.frame_colorization(fr, IB, last, fB, *sd, temperature=1e-10)
            dt = time.perf_counter() - t0
            last = torch.cat((fr[:, 0:1], ab), 1)
            if i >= 1:
                times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_timed} timed + 1 warm-up 216x384 frames of the same synthetic clip, oracle "
                      f"frame_colorization (reference op-for-op, exemplar side recomputed per frame as "
                      f"the reference does), torch CPU fp32, {cores} threads of {avail} available, median {med * 1e3:.0f} ms/frame"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hw", default="216x384",
                    help="frame size HxW; the default is BASELINE configs[1] (the metric's configuration), "
                         "432x768 is configs[3] (information only: no CPU baseline, traffic not re-measured)")
    ap.add_argument("--lookahead", type=int, default=2,
                    help="frames whose front end runs ahead on side HIP streams (0 = per-frame calls on one stream)")
    ap.add_argument("--no-autotune", action="store_true", help="use the static tile cost model instead of first-use timing")
    ap.add_argument("--corr", choices=["fp32", "bf16"], default="fp32",
                    help="bf16 = BASELINE configs[4]: bf16 MFMA candidate filter + exact fp32 re-scoring")
    ap.add_argument("--no-exemplar-cache", action="store_true",
                    help="recompute the exemplar side of WarpNet every frame, as the reference does")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    global H, W, P, CORR_FLOPS, CORR_BYTES, PATH_FLOPS, CORR_TRAFFIC_BYTES
    if args.hw != "216x384":
        H, W = (int(v) for v in args.hw.lower().split("x"))
        assert H % 16 == 0 and W % 16 == 0, "--hw: multiples of 16"
        scale = (H * W) / (216.0 * 384.0)
        P = (H // 4) * (W // 4)
        CORR_FLOPS = 2.0 * P * P * C + 2.0 * P * P * 3
        CORR_BYTES = 4.0 * (2 * P * C + 3 * P + 3 * P + P)
        PATH_FLOPS = (348.4e9 - 13.92e9) * scale + CORR_FLOPS     # convolutions scale with the pixels
        CORR_TRAFFIC_BYTES = None
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
    if args.gpus != world:
        log(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}")

    from dvc_amd import ops, synth
    from dvc_amd.frame import ClipColorizer
    from dvc_amd.parallel import broadcast_exemplar

    ops.set_autotune(not args.no_autotune)   # like the reference's cudnn.benchmark = True (test.py:140)
    nets, sd = build_nets(device)
    nets[1].corr_precision = args.corr
    cc = ClipColorizer(*nets, temperature=1e-10, cache_exemplar=not args.no_exemplar_cache)
    # exemplar: prepared on rank 0, shared once with every rank (RCCL broadcast over xGMI)
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(device)
    broadcast_exemplar(cc, IB if rank == 0 else None, (1, 3, H, W), device, src=0)

    K, Wm = args.steps, args.warmup
    # this rank's contiguous chunk of the clip: warm-up frames then K timed frames (resident in HBM)
    base = synth.FRAME_SEED0 + rank * (K + Wm)
    frames = [synth.synth_lab(base + i, H, W).to(device) for i in range(K + Wm)]
    last = torch.zeros(1, 3, H, W, device=device)

    def step(i, last):
        ab, _ = cc.frame(frames[i], last)
        return torch.cat((frames[i][:, 0:1], ab), dim=1)      # test.py:96

    for i in range(Wm):
        last = step(i, last)
    if args.lookahead > 0:
        # second untimed pass over the warm-up frames through the clip driver: the side streams' memory
        # pools (and nothing else) are still cold after the per-frame pass above
        cc.clip(frames[:Wm], lookahead=args.lookahead)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.lookahead > 0:
        # the clip driver: front end (VGG19 + WarpNet + correlation) of frames t+1.. on side HIP streams while
        # this stream runs the ColorVidNet recurrence; bit-identical to the per-frame loop below
        cc.clip(frames[Wm:Wm + K], last=last, lookahead=args.lookahead)
        last_timed = cc.last_lab
    else:
        last_timed = last
        for i in range(Wm, Wm + K):
            last_timed = step(i, last_timed)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # the same K frames through the reference's per-frame API (frame_colorization called frame by frame, no
    # look-ahead possible): reported next to `value`, and checked to give the same predictions
    seq_fps = None
    if args.lookahead > 0 and rank == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        last_seq = last
        for i in range(Wm, Wm + K):
            last_seq = step(i, last_seq)
        torch.cuda.synchronize()
        seq_fps = K / (time.perf_counter() - t1)
        assert torch.equal(last_seq, last_timed), "pipelined clip driver != per-frame loop"
    last = last_timed
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert torch.isfinite(last).all(), "non-finite output"
    fps = n_gpus * K / elapsed

    # ---- roofline of the north-star kernel: HIP events on the launch stream, same inputs
    roof = None
    if rank == 0:
        g = torch.Generator().manual_seed(1)
        th = ops.corr_prepare(torch.randn(1, C, P, generator=g).to(device))
        ph = ops.corr_prepare(torch.randn(1, C, P, generator=g).to(device))
        bl = torch.randn(1, 3, P, generator=g).to(device)
        for _ in range(3):
            ops.corr_fwd(th, ph, bl, 1e-10, H // 4, W // 4)
        reps = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                      # ops launch on torch's current stream, so these events see them
        for _ in range(reps):
            ops.corr_fwd(th, ph, bl, 1e-10, H // 4, W // 4)
        e1.record()
        torch.cuda.synchronize()
        t_corr = e0.elapsed_time(e1) * 1e-3 / reps      # fused kernel + its (tiny) merge kernel
        achieved = CORR_FLOPS / t_corr / 1e12
        roof = {"kernel": "corr_fwd_kernel (+corr_merge_kernel)", "bound": "mfma",
                "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                # HBM bytes per launch from the separate rocprofv3 --pmc passes (2 x FETCH_SIZE + WRITE_SIZE,
                # gfx950 correction per MI355X_MICROARCH.md); measured offline, see profiles/*_pmc_summary.md
                "traffic": CORR_TRAFFIC_BYTES, "traffic_source": CORR_TRAFFIC_SOURCE,
                "avg_launch_us": round(t_corr * 1e6, 2),
                "hbm_view": {"achieved": round(CORR_BYTES / t_corr / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": round(CORR_BYTES / t_corr / 1e9 / PEAK_HBM_GBS, 5),
                             "note": "fused kernel never materialises the PxP affinity; compulsory bytes "
                                     "10.76 MB/frame make it MFMA-bound, not HBM-bound (SURVEY.md 8d)"},
                "whole_path": {"achieved": round(PATH_FLOPS * fps / n_gpus / 1e12, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(PATH_FLOPS * fps / n_gpus / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}}

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd)

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
                       "conv_tile_choice": "static cost model" if args.no_autotune else
                       "autotuned on first use during warm-up (cf. cudnn.benchmark=True, test.py:140)",
                       "frames_per_gpu": K, "parallelism": f"frame-chunks x{n_gpus}",
                       "clip_driver": "per-frame calls, one stream" if args.lookahead <= 0 else
                       f"ClipColorizer.clip: front end of the next {args.lookahead} frames on side HIP streams, "
                       "ColorVidNet recurrence on the main stream (bit-identical to per-frame calls)",
                       "per_frame_api_frames_per_s": None if seq_fps is None else round(seq_fps, 3)},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
