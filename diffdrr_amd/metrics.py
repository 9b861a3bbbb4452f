"""Image similarity used by the registration / sweep paths.

Only normalised cross-correlation is on the path BASELINE.json's configs 4-5
exercise (reference ``diffdrr/metrics.py:16-63``); gradient-NCC, mutual
information and the geodesic pose metrics are out of scope.
"""
from __future__ import annotations

import torch

from . import ops


class _NCCFn(torch.autograd.Function):
    """Whole-image NCC of single-channel image pairs as one kernel each way
    (ddrr_ncc_forward / ddrr_ncc_backward).  x1 (B or 1, N), x2 (B, N) -> (B,)."""

    @staticmethod
    def forward(ctx, x1, x2, eps):
        out, stats = ops.ncc_forward(x1, x2, eps)
        ctx.save_for_backward(x1, x2, stats)
        return out

    @staticmethod
    def backward(ctx, g):
        x1, x2, stats = ctx.saved_tensors
        need1, need2 = ctx.needs_input_grad[:2]
        shared = x1.shape[0] == 1 and x2.shape[0] != 1
        if need1 and shared:
            # a fixed image shared by the batch: its gradient is a sum over the batch;
            # rare (the fixed image is data), so take the general route
            g1, g2 = ops.ncc_backward(x1.expand_as(x2).contiguous(), x2, stats, g, True, need2)
            return g1.sum(0, keepdim=True), g2, None
        g1, g2 = ops.ncc_backward(x1, x2, stats, g, need1, need2)
        return g1, g2, None


def to_patches(x, patch_size):
    """Every ``patch_size`` x ``patch_size`` window (stride 1) becomes one channel
    whose spatial extent is the window, so that :meth:`norm` z-scores each window
    on its own (local NCC), as in reference metrics.py:16-18."""
    x = x.unfold(2, patch_size, 1).unfold(3, patch_size, 1)  # b c h' w' p1 p2
    b, c, h, w, p1, p2 = x.shape
    return x.reshape(b, c * h * w, p1, p2)


class NormalizedCrossCorrelation2d(torch.nn.Module):
    """Mean of the product of the two z-scored images (global or patch-wise)."""

    def __init__(self, patch_size=None, eps=1e-5):
        super().__init__()
        self.patch_size = patch_size
        self.eps = eps

    def forward(self, x1, x2):
        if self.patch_size is not None:
            x1 = to_patches(x1, self.patch_size)
            x2 = to_patches(x2, self.patch_size)
        assert x1.shape == x2.shape, "Input images must be the same size"
        _, c, h, w = x1.shape
        if (self.patch_size is None and c == 1 and ops.on_device(x2) and x1.device == x2.device
                and x1.dtype == x2.dtype == torch.float32):
            # one fused kernel per direction; an `expand`ed fixed image is read once per pose
            # from the same memory instead of being materialised
            b = x2.shape[0]
            shared = b > 1 and x1.stride(0) == 0
            a = (x1[:1] if shared else x1).reshape(1 if shared else b, h * w)
            return _NCCFn.apply(a, x2.reshape(b, h * w), self.eps)
        score = (self.norm(x1) * self.norm(x2)).flatten(1).sum(1)
        return score / (c * h * w)

    def norm(self, x):
        mu = x.mean(dim=[-1, -2], keepdim=True)
        var = x.var(dim=[-1, -2], keepdim=True, correction=0) + self.eps
        return (x - mu) / var.sqrt()


class MultiscaleNormalizedCrossCorrelation2d(torch.nn.Module):
    """Weighted sum of NCCs at several patch sizes (``None`` = whole image)."""

    def __init__(self, patch_sizes=[None], patch_weights=[1.0], eps=1e-5):
        super().__init__()
        assert len(patch_sizes) == len(patch_weights), "Each scale must have a weight"
        self.nccs = [NormalizedCrossCorrelation2d(p, eps) for p in patch_sizes]
        self.patch_weights = patch_weights

    def forward(self, x1, x2):
        return sum(w * ncc(x1, x2) for w, ncc in zip(self.patch_weights, self.nccs))
