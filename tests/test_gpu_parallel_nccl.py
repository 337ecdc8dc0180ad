"""GPU, RCCL: dvc_amd.parallel with the REAL ClipColorizer over torch.distributed's `nccl` backend (= RCCL on
ROCm), one process per GPU, world = min(2, visible GPUs).  On a 1-GPU box this still executes the whole protocol
(process group, broadcasts, chunking, all_gather) with device tensors on one rank; with >= 2 GPUs the exemplar
cache really crosses xGMI.  Every rank's chunk must equal a single-process run of the same frames started from
I_last = 0, for both exemplar-cache layouts (fp32 and the bf16 candidate filter) — the protocol test, 48x80
— and, at the BASELINE configs[2] / configs[4] geometry (216x384, SURVEY §8(d) C3's seeds), every rank's chunk is compared
with THE ORACLE's chunk (tests/c3_common.py: the reference recurrence started from I_last = 0, with the tie-break matching
of near-tie rows), not with a second HIP run."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")
H, W, NF = 48, 80, 5


def _worker(rank, world, port, precision, q):
    try:
        for p in (PKG, ROOT):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import contextlib
        import io
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        from dvc_amd import parallel, synth
        from dvc_amd.frame import ClipColorizer
        from models.ColorVidNet import ColorVidNet
        from models.NonlocalNet import VGG19_pytorch, WarpNet
        sd = (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0, contractive=True))

        def build():
            with contextlib.redirect_stdout(io.StringIO()):
                nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
            for m, s in zip(nets, sd):
                m.load_state_dict(s)
                m.eval().to(dev)
            nets[1].corr_precision = precision
            return ClipColorizer(*nets, temperature=1e-10)

        frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W) for i in range(NF)]
        IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(dev)
        cc = build()
        lo, hi, outs, full = parallel.colorize_clip_sharded(cc, frames, IB if rank == 0 else None, dev, gather=True)
        torch.cuda.synchronize()
        ref = build()
        ref.set_exemplar(IB)
        want = ref.clip([f.to(dev) for f in frames[lo:hi]])
        ok_local = len(outs) == hi - lo and all(torch.equal(a, b) for a, b in zip(outs, want))
        ok_cache = all(torch.equal(a, b) for a, b in zip(cc.exemplar_cache_tensors(), ref.exemplar_cache_tensors()))
        ok_recv = rank == 0 or cc.features_B is None        # non-src ranks received the cache, did not compute it
        ok_full = len(full) == NF and all(torch.equal(full[lo + i], outs[i]) for i in range(hi - lo))
        q.put((rank, lo, hi, bool(ok_local), bool(ok_cache), bool(ok_recv), bool(ok_full), ""))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent instead of a bare exit code
        import traceback
        q.put((rank, -1, -1, False, False, False, False, traceback.format_exc()))
        raise


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_sharded_clip_nccl(precision):
    world = min(2, torch.cuda.device_count())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 150) + (1 if precision == "bf16" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, precision, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from dvc_amd.parallel import chunk_bounds
    assert [(r[1], r[2]) for r in res] == [chunk_bounds(NF, world, r) for r in range(world)]
    for r in res:
        assert all(r[3:7]), r


def _worker_oracle(rank, world, port, precision, nf, q):
    """configs[2] / configs[4] geometry: this rank's chunk of seeds 1000.. at 216x384 through colorize_clip_sharded (exemplar
    side computed on rank 0, received over RCCL elsewhere) against the oracle's run of THAT chunk."""
    try:
        for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import contextlib
        import io
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        import c3_common as C
        from dvc_amd import parallel, synth
        from dvc_amd.frame import ClipColorizer
        from models.ColorVidNet import ColorVidNet
        from models.NonlocalNet import VGG19_pytorch, WarpNet
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(32, avail) // world))
        HH, WW, T = 216, 384, 1e-10
        sd = (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0, contractive=True))
        with contextlib.redirect_stdout(io.StringIO()):
            nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
        for m, s_ in zip(nets, sd):
            m.load_state_dict(s_)
            m.eval().to(dev)
        nets[1].corr_precision = precision
        cc = ClipColorizer(*nets, temperature=T)
        frames = [synth.synth_lab(synth.FRAME_SEED0 + i, HH, WW) for i in range(nf)]
        IB = synth.synth_lab(synth.EXEMPLAR_SEED, HH, WW)
        lo, hi, outs, _ = parallel.colorize_clip_sharded(cc, frames, IB.to(dev) if rank == 0 else None, dev, gather=False)
        torch.cuda.synchronize()
        hip_fronts = [C.hip_front(nets[0], nets[1], cc, f.to(dev), T) for f in frames[lo:hi]]
        phi = C.oracle_exemplar(sd, IB)
        fronts = [C.oracle_front(sd, IB, phi, f, T) for f in frames[lo:hi]]
        want, stats = C.matched_oracle_chunk(sd, IB, frames[lo:hi], fronts, hip_fronts)
        errs = [(g.cpu() - w_).abs().max().item() for g, w_ in zip(outs, want)]
        ok = (len(outs) == hi - lo and all(e <= 1e-3 for e in errs) and all(gp < 1e-5 for st in stats for gp in st["gaps"])
              and all(st["sim_err"] < 1e-5 for st in stats))
        q.put((rank, lo, hi, bool(ok), errs, [st["flipped"] for st in stats], ""))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, -1, -1, False, [], [], traceback.format_exc()))
        raise


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_sharded_clip_nccl_against_the_oracle_216x384(precision):
    """16 frames (seeds 1000..1015, half of them with a near-tie row) at 216x384 sharded over min(2, visible GPUs) ranks: every
    rank's chunk within the north-star 1e-3 of the tie-break-matched oracle recurrence on that chunk."""
    world = min(2, torch.cuda.device_count())
    nf = 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29960 + (os.getpid() % 30) + (1 if precision == "bf16" else 0)
    procs = [ctx.Process(target=_worker_oracle, args=(r, world, port, precision, nf, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from dvc_amd.parallel import chunk_bounds
    rep = os.path.join(ROOT, "gpurun_out", "test_report.txt")
    os.makedirs(os.path.dirname(rep), exist_ok=True)
    with open(rep, "a") as f:
        for r in res:
            f.write(f"sharded nccl vs oracle {precision} world={world} rank {r[0]} frames [{r[1]},{r[2]}): ab max error per frame vs the tie-break-"
                    f"matched oracle {['%.1e' % e for e in r[4]]}; flipped rows per frame {r[5]}\n")
    assert [(r[1], r[2]) for r in res] == [chunk_bounds(nf, world, r) for r in range(world)]
    for r in res:
        assert r[3], r
