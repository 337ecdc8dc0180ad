"""CPU restatement of the reference's contextual losses — TEST INFRASTRUCTURE ONLY (imported by tests/ and by
oracle/pin_contextual.py; the product package never imports anything under oracle/).

Follows /root/reference/models/ContextualLoss.py op for op:
  contextual_loss_forward   ContextualLoss_forward.forward   :97-126  (row maxima: max over dim=-1, mean over dim=1)
  contextual_loss           ContextualLoss.forward           :38-77   (column maxima: max over dim=1, mean over dim=-1)
with `feature_normalize` of utils/util.py:155-158.  Pinned bit-exact (float32, values and autograd gradients) against the
unmodified reference module by oracle/pin_contextual.py, whose outputs are stored in tests/golden/contextual_*.npz.
dtype-generic: the GPU parity tests run it in float64 as the truth.
"""
import sys

import torch


def feature_normalize(feature_in):                     # utils/util.py:155-158
    feature_in_norm = torch.norm(feature_in, 2, 1, keepdim=True) + sys.float_info.epsilon
    return torch.div(feature_in, feature_in_norm)


def _affinity(X_features, Y_features, h, feature_centering):          # ContextualLoss.py:43-72 == :102-121
    batch_size = X_features.shape[0]
    feature_depth = X_features.shape[1]
    if feature_centering:
        mu = Y_features.view(batch_size, feature_depth, -1).mean(dim=-1).unsqueeze(dim=-1).unsqueeze(dim=-1)
        X_features = X_features - mu
        Y_features = Y_features - mu
    X_features = feature_normalize(X_features).view(batch_size, feature_depth, -1)
    Y_features = feature_normalize(Y_features).view(batch_size, feature_depth, -1)
    X_features_permute = X_features.permute(0, 2, 1)
    d = 1 - torch.matmul(X_features_permute, Y_features)
    d_norm = d / (torch.min(d, dim=-1, keepdim=True)[0] + 1e-5)
    w = torch.exp((1 - d_norm) / h)
    A_ij = w / torch.sum(w, dim=-1, keepdim=True)
    return A_ij


def contextual_loss_forward(X_features, Y_features, h=0.1, feature_centering=True):
    A_ij = _affinity(X_features, Y_features, h, feature_centering)
    CX = torch.mean(torch.max(A_ij, dim=-1)[0], dim=1)                 # :125
    return -torch.log(CX)


def contextual_loss(X_features, Y_features, h=0.1, feature_centering=True):
    A_ij = _affinity(X_features, Y_features, h, feature_centering)
    CX = torch.mean(torch.max(A_ij, dim=1)[0], dim=-1)                 # :76
    return -torch.log(CX)


def synth_features(seed, B, C, H, W, correlated=0.35):
    """Feature pair for the tests: Y random non-negative (post-ReLU like VGG taps), X = a noisy, spatially permuted mixture
    of Y's columns plus fresh noise — the rows of the affinity have a best match without being one-hot (losses of 0.3 .. 2.5
    at the reference's bandwidth h = 0.1)."""
    g = torch.Generator().manual_seed(seed)
    Y = torch.relu(torch.randn(B, C, H, W, generator=g) + 0.3)
    perm = torch.randperm(H * W, generator=g)
    Xs = Y.view(B, C, -1)[:, :, perm].view(B, C, H, W)
    X = torch.relu(correlated * Xs + (1 - correlated) * torch.randn(B, C, H, W, generator=g) + 0.1)
    return X, Y
