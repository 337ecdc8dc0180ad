"""Multi-GPU: one process per GPU, frames of a clip sharded in contiguous chunks.

What shards: VGG(A) + WarpNet(A) + correlation are frame-independent; ColorVidNet of
frame t consumes frame t-1's prediction (test.py:96), a strict chain inside a clip.  The partition
that keeps reference semantics is therefore *contiguous chunks treated as independent clips*: rank r
colourises frames [lo_r, hi_r) starting from I_last = 0 exactly like the reference's first frame
(test.py:76-80).  This equals the reference run on each chunk; it is NOT bit-identical to one long
sequential run at the chunk boundaries (first frame of a chunk sees zeros instead of the previous
prediction).

Collectives (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" in CPU tests):
  * ONE broadcast per clip of the exemplar, a single flat byte buffer from rank 0: IB_lab (1 MB) and, when the
    exemplar side is cached, its products phi (256 x P fp32 = 5.3 MB at 216x384) and the pooled Lab (62 KB);
  * an optional all_gather of the ab predictions (663 KB/frame) when one rank needs the whole clip.
There is no collective in the per-frame path.
"""
import torch
import torch.distributed as dist


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def chunk_bounds(n_frames, world, rank):
    """Contiguous, balanced chunks: the first (n_frames % world) ranks get one extra frame."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_exemplar(cc, IB_lab, shape, device, src=0):
    """Give every rank's ClipColorizer the same exemplar state, computing the exemplar side once.

    `cc` needs: .cache_exemplar, .set_exemplar(IB_lab), .IB_lab, and — when cache_exemplar is on —
    .exemplar_cache_spec(shape) / .exemplar_cache_tensors() / .load_exemplar_cache(IB, tensors)
    (fp32 cache: phi + pooled Lab, 5.4 MB at 216x384; bf16 candidate-filter cache: phi in fp32 and bf16 +
    pooled Lab, 8 MB).  ONE collective per clip: the exemplar Lab and the cache tensors travel as one flat byte
    buffer (every tensor at a 16-byte aligned offset; bytes, because RCCL has no int16 for the bf16 bit patterns)
    from rank `src` to all — a ring/tree broadcast pays its per-link latency once instead of three or four times."""
    world, rank = _world()
    if not (dist.is_available() and dist.is_initialized()):
        cc.set_exemplar(IB_lab)
        return
    spec = [(tuple(shape), torch.float32)]
    if cc.cache_exemplar:
        spec += [(tuple(s), dt) for s, dt in cc.exemplar_cache_spec(shape)]
    offs, total = flat_layout(spec)
    flat = torch.empty(total, device=device, dtype=torch.uint8)
    views = [flat[o:o + _nbytes(s, dt)].view(dt).view(s) for o, (s, dt) in zip(offs, spec)]
    if rank == src:
        IB = IB_lab.contiguous().float()
        src_tensors = [IB]
        if cc.cache_exemplar:
            cc.set_exemplar(IB)
            src_tensors += cc.exemplar_cache_tensors()
        for v, t in zip(views, src_tensors):
            v.copy_(t)
    dist.broadcast(flat, src)
    if rank == src:
        if not cc.cache_exemplar:
            cc.set_exemplar(IB)
        return
    if not cc.cache_exemplar:
        cc.set_exemplar(views[0])    # every rank prepares (and re-prepares per frame) the exemplar side itself
        return
    cc.load_exemplar_cache(views[0], views[1:])


def _nbytes(shape, dtype):
    n = torch.empty((), dtype=dtype).element_size()
    for d in shape:
        n *= int(d)
    return n


def flat_layout(spec):
    """[(shape, dtype)] -> (byte offset of every tensor, total bytes): consecutive, each start rounded up to 16 bytes."""
    offs, o = [], 0
    for s, dt in spec:
        offs.append(o)
        o = (o + _nbytes(s, dt) + 15) // 16 * 16
    return offs, o


def colorize_clip_sharded(cc, frames_lab, IB_lab, device, gather=True, src=0):
    """Colourise `frames_lab` (list of 1x3xHxW Lab tensors, same list on every rank or at least this
    rank's chunk valid) across all ranks.  Returns (lo, hi, local_outputs, gathered_or_None)."""
    world, rank = _world()
    n = len(frames_lab)
    lo, hi = chunk_bounds(n, world, rank)
    shape = tuple(frames_lab[lo if hi > lo else 0].shape)
    broadcast_exemplar(cc, IB_lab if rank == src else None, shape, device, src=src)
    outs = cc.clip([f.to(device) for f in frames_lab[lo:hi]]) if hi > lo else []
    if not gather or world == 1:
        return lo, hi, outs, (outs if world == 1 else None)
    max_chunk = -(-n // world)
    pad = torch.zeros((max_chunk, 2) + shape[2:], device=device, dtype=torch.float32)
    for i, o in enumerate(outs):
        pad[i] = o[0]
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    full = []
    for r in range(world):
        l, h = chunk_bounds(n, world, r)
        full.extend(parts[r][i:i + 1] for i in range(h - l))
    return lo, hi, outs, full
