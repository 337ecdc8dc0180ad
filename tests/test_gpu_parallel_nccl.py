"""GPU, RCCL: dvc_amd.parallel with the REAL ClipColorizer over torch.distributed's `nccl` backend (= RCCL on
ROCm), one process per GPU, world = min(2, visible GPUs).  On a 1-GPU box this still executes the whole protocol
(process group, broadcasts, chunking, all_gather) with device tensors on one rank; with >= 2 GPUs the exemplar
cache really crosses xGMI.  Every rank's chunk must equal a single-process run of the same frames started from
I_last = 0, for both exemplar-cache layouts (fp32 and the bf16 candidate filter)."""
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
