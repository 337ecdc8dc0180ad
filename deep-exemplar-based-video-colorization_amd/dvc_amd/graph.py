"""hipGraph replay of the per-frame launch sequences (the loop they serve is
/root/reference/test.py:68-96).

One frame is ~120 kernel launches issued from one Python thread (1.65 ms of host time per 2.4 ms frame, tools/
host_issue_probe.py): the front end (gray2rgb -> VGG19 -> feature_normalize -> WarpNet heads / trunk -> theta -> fused
correlation, ~75 launches) and the ColorVidNet chain (~45).  Both are static per frame geometry — same kernels, same grid
sizes, same buffers — so each is captured ONCE into a hipGraph (stream capture of the very calls the eager path makes:
every C-ABI entry point only enqueues on the stream it is given, include/dvc_hip.h) and replayed per frame with one
hipGraphLaunch.  Same kernels, same arguments, same order: the replayed results are bit-identical to the eager ones
(tests/test_gpu_nets.py::test_graph_replay_equals_eager).

PyTorch is the plumbing here as everywhere in this package: `torch.cuda.CUDAGraph` IS hipStreamBeginCapture /
hipGraphInstantiate / hipGraphLaunch on ROCm, and its capture-time allocator pool is what gives every intermediate tensor
of the sequence a fixed address (the "static arena").  What a capture must not contain is kept out by construction:
  * weight packing (it synchronises: nets._PackCache) and conv autotuning (it times launches) happen in an eager warm-up
    run of the same function right before the capture; `nets.pack_epoch()` tells a later caller that weights were repacked
    (load_state_dict, .cuda(), another conv algorithm) and the sequence has to be captured again;
  * scratch memory: a captured sequence bakes its workspace addresses in and may be replayed on any stream, so it gets
    private workspaces (ops.workspace_scope) instead of the per-stream ones eager launches use.
"""
import torch

from . import nets, ops


def _epoch():
    """Everything that decides WHICH launches a sequence consists of."""
    return (nets.pack_epoch(), ops.conv_algo(), ops.fuse_reduce(), ops.pool_fusion(), ops.dual_conv_enabled(), ops.autotune_enabled(),
            ops.fold_merge(), ops.direct_layers(), ops.batch_plan_enabled(), ops.group_heads(), ops.ws_conv_enabled(), ops.gray_fusion())


class CapturedSequence:
    """`fn()` (no arguments: it closes over its static input tensors) captured on `stream`; `.out` is whatever `fn`
    returned at capture time (tensors at fixed addresses, overwritten by every replay)."""

    def __init__(self, fn, stream):
        self.workspaces = {}
        self.graph = torch.cuda.CUDAGraph()
        stream.wait_stream(torch.cuda.current_stream())     # whatever `fn` reads was produced on the caller's stream
        with ops.workspace_scope(self.workspaces):
            with torch.cuda.stream(stream):
                fn()                                  # eager warm-up: packs weights, tunes convolutions, sizes workspaces
            stream.synchronize()
            self.epoch = _epoch()
            with torch.cuda.graph(self.graph, stream=stream):
                self.out = fn()
        if _epoch() != self.epoch:
            raise RuntimeError("dvc_amd.graph: a weight was packed during stream capture (warm-up did not cover the sequence)")

    def stale(self):
        return _epoch() != self.epoch

    def replay(self):
        """One hipGraphLaunch on the current stream."""
        self.graph.replay()
