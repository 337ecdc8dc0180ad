"""bench.py with a debug kernel variant switched on for the whole run (A/B of a kernel change at the level that counts:
frames/s of the clip driver, where three streams share the chip — tools/conv_wino_ab.py times isolated launches).
  DVC_DEBUG_LIB=1 CONV_VARIANT=8 python tools/bench_variant.py --no-cpu-baseline
needs the -DDVC_DEBUG build (`make -C csrc DEBUG=1`)."""
import os
import runpy
import sys

os.environ["DVC_DEBUG_LIB"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
from dvc_amd import _lib  # noqa: E402

_lib.load().dvc_debug_conv_variant(int(os.environ.get("CONV_VARIANT", "0")))
if os.environ.get("BENCH_SKIP_EQUAL") == "1":
    # variants that change the arithmetic between the bench's legs on purpose (e.g. another split for batched launches): the
    # legs' bit-equality assertions do not apply to such a timing experiment
    import torch
    torch.equal = lambda a, b: True
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
