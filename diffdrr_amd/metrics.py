"""Image similarities of the registration / sweep paths.

Normalised cross-correlation -- whole-image, patch-wise, multiscale -- and its gradient
variant (reference ``diffdrr/metrics.py:16-104``); mutual information and the geodesic pose
metrics are out of scope.  Pinned to ``tests/golden/metrics.npz``, values
and autograd gradients of the unmodified reference.
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


class _PatchNCCFn(torch.autograd.Function):
    """Patch-wise NCC (``patch_size = p``) of single-channel image pairs without the reference's
    (B, windows, p, p) tensors: ddrr_ncc_patch_forward / ddrr_ncc_patch_backward.
    x1 (B or 1, H, W), x2 (B, H, W) -> (B,).  The gradient w.r.t. the moving image x2 is one launch;
    one w.r.t. x1 (the fixed image is data: rare) is the same two launches with the roles swapped --
    the similarity is symmetric."""

    @staticmethod
    def forward(ctx, x1, x2, p, eps):
        out, coef = ops.ncc_patch_forward(x1, x2, p, eps, want_coef=ctx.needs_input_grad[1])
        ctx.p, ctx.eps = p, eps
        ctx.save_for_backward(x1, x2, coef)
        return out

    @staticmethod
    def backward(ctx, g):
        x1, x2, coef = ctx.saved_tensors
        g1 = g2 = None
        if ctx.needs_input_grad[1]:
            g2 = ops.ncc_patch_backward(x1, x2, coef, g, ctx.p)
        if ctx.needs_input_grad[0]:
            a = x1.expand_as(x2).contiguous()
            _, coef1 = ops.ncc_patch_forward(x2, a, ctx.p, ctx.eps)
            g1 = ops.ncc_patch_backward(x2, a, coef1, g, ctx.p)
            if x1.shape[0] == 1 and x2.shape[0] != 1:
                g1 = g1.sum(0, keepdim=True)
        return g1, g2, None, None


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
        if (self.patch_size is not None and not getattr(self, "_no_patch_kernel", False)
                and x1.dim() == 4 and x1.shape == x2.shape and ops.on_device(x2)
                and x1.device == x2.device and x1.dtype == x2.dtype == torch.float32
                and 1 <= int(self.patch_size) <= min(x2.shape[-2], x2.shape[-1], 64)):
            # local NCC without `to_patches`: every window z-scored in LDS; the channels are
            # independent images (the reference's mean over (c h' w') windows = the mean over channels
            # of the per-channel means); an `expand`ed fixed image is read once, in place
            b, c, h, w = x2.shape
            shared = b > 1 and x1.stride(0) == 0 and c == 1
            a = (x1[:1] if shared else x1).reshape(1 if shared else b * c, h, w)
            val = _PatchNCCFn.apply(a, x2.reshape(b * c, h, w), int(self.patch_size), self.eps)
            return val.reshape(b, c).mean(dim=1) if c > 1 else val
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


class _SobelFn(torch.autograd.Function):
    """(B, 1, H, W) -> (B, 2, H, W) Sobel responses, zero padding: ddrr_sobel_forward / its
    adjoint ddrr_sobel_backward (reference metrics.py:69-94, torch.nn.Conv2d(1, 2, 3, padding=1))."""

    @staticmethod
    def forward(ctx, x):
        return ops.sobel_forward(x[:, 0])

    @staticmethod
    def backward(ctx, g):
        return ops.sobel_backward(g).unsqueeze(1)


class _BlurSobelFn(torch.autograd.Function):
    """(B, 1, H, W) -> (B, 2, H, W): Gaussian blur (reflect padding) + Sobel pair, one launch each way:
    ddrr_blur_sobel_forward / ddrr_blur_sobel_backward (reference metrics.py:88-93)."""

    @staticmethod
    def forward(ctx, x, taps):
        ctx.taps = taps
        return ops.blur_sobel_forward(x[:, 0], taps)

    @staticmethod
    def backward(ctx, g):
        return ops.blur_sobel_backward(g, ctx.taps).unsqueeze(1), None


def gaussian_taps(kernel_size, sigma, dtype=torch.float32, device=None):
    """The 1-D taps of torchvision's gaussian_blur (see :func:`gaussian_blur`)."""
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size, dtype=dtype, device=device)
    k1 = torch.exp(-0.5 * (x / sigma).pow(2))
    return k1 / k1.sum()


def gaussian_blur(img, kernel_size, sigma):
    """torchvision.transforms.functional.gaussian_blur as the reference's ``Sobel`` calls it
    (metrics.py:66, 88-92; torchvision is a third-party dependency that is neither vendored by
    the reference nor installed here: restated from its published algorithm): taps exp(-x^2 / 2 sigma^2) on
    linspace(-(k-1)/2, (k-1)/2, k), normalised, separable, REFLECT padding by k // 2.  A few tensor
    ops on (B, 1, H, W) images; differentiable by autograd."""
    k1 = gaussian_taps(kernel_size, sigma, img.dtype, img.device)
    c = img.shape[-3]
    kernel = torch.mm(k1[:, None], k1[None, :]).expand(c, 1, kernel_size, kernel_size)
    pad = [kernel_size // 2] * 4
    return torch.nn.functional.conv2d(torch.nn.functional.pad(img, pad, mode="reflect"), kernel, groups=c)


class Sobel(torch.nn.Module):
    """Optional Gaussian blur, then the x / y Sobel responses (reference metrics.py:69-94)."""

    def __init__(self, sigma: float):
        super().__init__()
        self.sigma = sigma
        self._taps = {}  # device -> the blur's taps there (computed on the host as torchvision does)

    def forward(self, img):
        x = img
        if self.sigma > 0:
            k = int(6 * self.sigma + 1) | 1
            if (img.dim() == 4 and img.shape[1] == 1 and ops.on_device(img) and img.dtype == torch.float32
                    and k <= 31 and k // 2 < min(img.shape[-2:]) and not getattr(self, "_no_blur_kernel", False)):
                key = (img.device, float(self.sigma))
                if key not in self._taps:
                    self._taps = {key: gaussian_taps(k, self.sigma).to(img.device)}
                return _BlurSobelFn.apply(img, self._taps[key])
            x = gaussian_blur(img, k, self.sigma)
        if (x.shape[1] == 1 and ops.on_device(x) and x.dtype == torch.float32):
            return _SobelFn.apply(x)
        Gx = torch.tensor([[1, 0, -1], [2, 0, -2], [1, 0, -1]], dtype=x.dtype, device=x.device)
        Gy = torch.tensor([[1, 2, 1], [0, 0, 0], [-1, -2, -1]], dtype=x.dtype, device=x.device)
        return torch.nn.functional.conv2d(x, torch.stack([Gx, Gy]).unsqueeze(1), padding=1)


class GradientNormalizedCrossCorrelation2d(NormalizedCrossCorrelation2d):
    """NCC between the image gradients of two batches of images (reference metrics.py:97-104):
    the Sobel pair by one kernel each way, the two channels' whole-image NCCs by the fused NCC
    kernels (their mean is the reference's sum over (c, h, w) / (c h w))."""

    def __init__(self, patch_size=None, sigma=1.0, **kwargs):
        super().__init__(patch_size, **kwargs)
        self.sobel = Sobel(sigma)

    def forward(self, x1, x2):
        g1, g2 = self.sobel(x1), self.sobel(x2)
        b, c, h, w = g2.shape
        if (self.patch_size is None and c == 2 and ops.on_device(g2) and g1.shape == g2.shape
                and g2.dtype == torch.float32):
            # every channel is z-scored on its own: 2 B single-channel pairs for the fused kernels
            val = _NCCFn.apply(g1.reshape(b * c, h * w), g2.reshape(b * c, h * w), self.eps)
            return val.reshape(b, c).mean(dim=1)
        return super().forward(g1, g2)


class MultiscaleNormalizedCrossCorrelation2d(torch.nn.Module):
    """Weighted sum of NCCs at several patch sizes (``None`` = whole image)."""

    def __init__(self, patch_sizes=[None], patch_weights=[1.0], eps=1e-5):
        super().__init__()
        assert len(patch_sizes) == len(patch_weights), "Each scale must have a weight"
        # (like the reference, metrics.py:53-55, the members keep the default eps)
        self.nccs = [NormalizedCrossCorrelation2d(p) for p in patch_sizes]
        self.patch_weights = patch_weights

    def forward(self, x1, x2):
        # sum_i w_i ncc_i (reference metrics.py:58-63: the weighted scores stacked and summed) with one
        # launch per scale: the first scaled, the others added with their weight as `alpha`
        total = None
        for w, ncc in zip(self.patch_weights, self.nccs):
            v = ncc(x1, x2)
            if total is None:
                total = v * w
            elif isinstance(w, (int, float)):
                total = torch.add(total, v, alpha=w)
            else:  # (a tensor weight)
                total = total + v * w
        return total
