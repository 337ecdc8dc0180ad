"""Experiment: how much does running two independent frame pipelines on two HIP streams raise the
aggregate frame rate?  (Upper bound for cross-frame pipelining inside one clip.)"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dvc_amd import ops, synth  # noqa: E402
from dvc_amd.frame import ClipColorizer  # noqa: E402

dev = torch.device("cuda:0")
H, W = 216, 384
nets, _ = bench.build_nets(dev)
ops.set_autotune(True)
K = 24
frames = [synth.synth_lab(1000 + i, H, W).to(dev) for i in range(K)]
IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(dev)


def run(nstreams):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    ccs = []
    for s in streams:
        with torch.cuda.stream(s):
            cc = ClipColorizer(*nets, temperature=1e-10)
            cc.set_exemplar(IB)
            ccs.append(cc)
    last = [torch.zeros_like(frames[0]) for _ in streams]
    torch.cuda.synchronize()

    def go(n):
        for i in range(n):
            for k, s in enumerate(streams):
                with torch.cuda.stream(s):
                    ab, _ = ccs[k].frame(frames[i], last[k])
                    last[k] = torch.cat((frames[i][:, 0:1], ab), dim=1)
    go(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return nstreams * K / dt


for n in (1, 2, 1, 2, 3):
    print(f"{n} stream(s): {run(n):.1f} frames/s aggregate")
