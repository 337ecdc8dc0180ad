"""CPU: the contextual-loss oracle (oracle/contextual_oracle.py) against the fixtures recorded from the UNMODIFIED reference
module by oracle/pin_contextual.py (values and autograd gradients, float32, single-threaded: bit for bit), and the module
surface of the drop-in `models.ContextualLoss`."""
import glob
import os

import numpy as np
import torch

from oracle import contextual_oracle as O


def test_contextual_oracle_reproduces_reference_fixtures(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "contextual_*.npz")))
    assert len(files) >= 3
    old = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for f in files:
            g = np.load(f)
            B, C, H, W = (int(v) for v in g["shape"])
            X, Y = O.synth_features(int(g["seed"]), B, C, H, W)
            for tag, fn in (("fwd", O.contextual_loss_forward), ("bwd", O.contextual_loss)):
                x = X.clone().requires_grad_(True)
                loss = fn(x, Y, h=float(g["h"]), feature_centering=bool(int(g["centre"])))
                loss.sum().backward()
                assert np.array_equal(loss.detach().numpy(), g[f"loss_{tag}"]), (f, tag)
                assert np.array_equal(x.grad.numpy(), g[f"dx_{tag}"]), (f, tag)
                assert loss.min().item() > 0.2          # not the degenerate one-hot regime
    finally:
        torch.set_num_threads(old)


def test_drop_in_module_exports_and_fails_loudly_on_cpu():
    import pytest
    from models.ContextualLoss import ContextualLoss, ContextualLoss_forward
    import dvc_amd.contextual as C
    assert ContextualLoss is C.ContextualLoss and ContextualLoss_forward is C.ContextualLoss_forward
    X, Y = O.synth_features(1, 1, 16, 4, 4)
    for m in (ContextualLoss(), ContextualLoss_forward()):
        assert list(m.parameters()) == [] and list(m.state_dict()) == []     # like the reference: no parameters
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m(X, Y)
