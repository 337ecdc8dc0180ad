"""CPU, world_size 2 over gloo: the frame-chunk sharding + exemplar broadcast + gather logic of
dvc_amd.parallel, driven with an oracle-backed stand-in for the HIP ClipColorizer (tests may use the
oracle; the product path never does).  Checks SURVEY.md §8(e): every rank's chunk equals the
single-process run of that chunk started from I_last = 0."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")

H, W, NF = 32, 48, 5


class OracleColorizer:
    """Same interface as dvc_amd.frame.ClipColorizer, CPU oracle inside (TEST ONLY)."""

    def __init__(self, sd, cache_exemplar):
        self.sd = sd
        self.cache_exemplar = cache_exemplar
        self.IB_lab = self.features_B = self.ex_cache = None
        self.n_set_exemplar = 0

    def set_exemplar(self, IB_lab):
        from oracle import dvc_oracle as O
        self.n_set_exemplar += 1
        self.IB_lab = IB_lab
        with torch.no_grad():
            self.features_B = O.exemplar_features(IB_lab, self.sd[0])
            if self.cache_exemplar:
                nB = [O.feature_normalize(t) for t in self.features_B[1:]]
                phi = O.corr_project(self.sd[1], "phi", O.warp_features(self.sd[1], *nB))
                blab = torch.nn.functional.avg_pool2d(IB_lab, 4)
                self.ex_cache = (phi, blab)

    def exemplar_cache_spec(self, s):
        return [((s[0], 256, (s[2] // 4) * (s[3] // 4)), torch.float32), ((s[0], 3, s[2] // 4, s[3] // 4), torch.float32)]

    def exemplar_cache_tensors(self):
        return [t.contiguous() for t in self.ex_cache]

    def load_exemplar_cache(self, IB_lab, tensors):
        self.IB_lab, self.features_B, self.ex_cache = IB_lab, None, tuple(tensors)

    def clip(self, frames):
        from oracle import dvc_oracle as O
        outs, last = [], None
        with torch.no_grad():
            if self.features_B is None:          # non-src rank that only received the cached products
                feats = O.exemplar_features(self.IB_lab, self.sd[0])
            else:
                feats = self.features_B
            for fr in frames:
                if last is None:
                    last = torch.zeros_like(fr)
                ab, _, _ = O.frame_colorization(fr, self.IB_lab, last, feats, *self.sd, temperature=1e-10)
                last = torch.cat((fr[:, 0:1], ab), 1)
                outs.append(ab)
        return outs


def _worker(rank, world, port, cache, q):
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvc_amd import parallel, synth
    sd = (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))
    frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W) for i in range(NF)]
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    cc = OracleColorizer(sd, cache)
    n_bcast = []
    real_broadcast = dist.broadcast
    dist.broadcast = lambda t, src, *a, **k: (n_bcast.append(t.numel() * t.element_size()), real_broadcast(t, src, *a, **k))[1]
    lo, hi, outs, full = parallel.colorize_clip_sharded(cc, frames, IB if rank == 0 else None,
                                                        torch.device("cpu"), gather=True)
    # single-process reference for this rank's chunk
    ref = OracleColorizer(sd, False)
    ref.set_exemplar(IB)
    want = ref.clip(frames[lo:hi])
    ok_local = all(torch.equal(a, b) for a, b in zip(outs, want)) and len(outs) == hi - lo
    dist.broadcast = real_broadcast
    ok_ex = torch.equal(cc.IB_lab, IB) and (not cache or rank == 0 or cc.n_set_exemplar == 0)
    ok_ex = ok_ex and len(n_bcast) == 1       # ONE collective per clip: exemplar Lab + cache tensors in one flat buffer
    ok_full = len(full) == NF and all(torch.equal(full[lo + i], outs[i]) for i in range(hi - lo))
    q.put((rank, lo, hi, ok_local, ok_ex, ok_full))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cache", [True, False])
def test_sharded_clip_world2_gloo(cache):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + (1 if cache else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cache, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 5)]
    for r in res:
        assert r[3] and r[4] and r[5], r


def test_flat_layout_is_aligned_and_dense():
    from dvc_amd.parallel import flat_layout
    spec = [((1, 3, 216, 384), torch.float32), ((1, 5184, 256), torch.int16), ((1, 3, 5), torch.float32), ((7,), torch.uint8)]
    offs, total = flat_layout(spec)
    assert offs[0] == 0 and all(o % 16 == 0 for o in offs) and total % 16 == 0
    sizes = [3 * 216 * 384 * 4, 5184 * 256 * 2, 60, 7]
    for o, n, nxt in zip(offs, sizes, offs[1:] + [total]):
        assert o + n <= nxt < o + n + 16


def test_chunk_bounds_cover_and_balance():
    from dvc_amd.parallel import chunk_bounds
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 4, 8):
            spans = [chunk_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_clipcolorizer_exemplar_cache_spec_and_roundtrip():
    """The real ClipColorizer's side of the broadcast protocol (shapes/dtypes a receiving rank allocates, flat
    tensor order, re-installation) for both cache layouts: fp32 (phi, pooled Lab) and the bf16 candidate filter
    ((phi fp32, phi bf16-as-int16), pooled Lab).  Host logic only: no kernel runs."""
    import contextlib
    import io
    from dvc_amd.frame import ClipColorizer
    from models.NonlocalNet import WarpNet
    with contextlib.redirect_stdout(io.StringIO()):
        warp = WarpNet(1)
    cc = ClipColorizer(None, warp, None, temperature=1e-10)
    shape = (1, 3, 216, 384)
    P = 54 * 96
    assert cc.exemplar_cache_spec(shape) == [((1, 256, P), torch.float32), ((1, 3, 54, 96), torch.float32)]
    assert cc.exemplar_cache_shapes(shape) == [(1, 256, P), (1, 3, 54, 96)]
    bufs = [torch.zeros(s, dtype=dt) for s, dt in cc.exemplar_cache_spec(shape)]
    cc.load_exemplar_cache(torch.zeros(shape), bufs)
    assert cc.ex_cache[0] is bufs[0] and cc.ex_cache[1] is bufs[1] and cc.features_B is None
    assert all(a is b or torch.equal(a, b) for a, b in zip(cc.exemplar_cache_tensors(), bufs))
    warp.corr_precision = "bf16"
    spec = cc.exemplar_cache_spec(shape)
    assert spec == [((1, P, 256), torch.float32), ((1, P, 256), torch.int16), ((1, 3, 54, 96), torch.float32)]
    bufs = [torch.zeros(s, dtype=dt) for s, dt in spec]
    cc.load_exemplar_cache(torch.zeros(shape), bufs)
    assert isinstance(cc.ex_cache[0], tuple) and cc.ex_cache[0][1].dtype == torch.int16
    assert [t.dtype for t in cc.exemplar_cache_tensors()] == [dt for _, dt in spec]
    cc.temperature = 0.01            # soft temperature: the bf16 filter is not used, fp32 layout again
    assert len(cc.exemplar_cache_spec(shape)) == 2
