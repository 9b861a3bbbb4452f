"""Shim: reference metrics.py:66 imports ``gaussian_blur`` at module level (``Sobel`` calls it
for sigma > 0).  torchvision is absent from this container; this is a restatement of the
published algorithm of ``torchvision.transforms.functional.gaussian_blur`` (torchvision 0.2x,
``_functional_tensor.gaussian_blur``): a separable Gaussian with taps exp(-x^2 / 2 sigma^2) on
x = linspace(-(k-1)/2, (k-1)/2, k), normalised to sum 1, applied as one depthwise 2-D convolution
after REFLECT padding by k // 2.  Test infrastructure only (fixture generation)."""
import torch
import torch.nn.functional as F


def _kernel1d(kernel_size: int, sigma: float, dtype, device):
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size, dtype=dtype, device=device)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(img, kernel_size, sigma=None):
    ks = [kernel_size, kernel_size] if isinstance(kernel_size, int) else list(kernel_size)
    sg = [sigma, sigma] if isinstance(sigma, (int, float)) else list(sigma)
    kx = _kernel1d(ks[0], float(sg[0]), img.dtype, img.device)
    ky = _kernel1d(ks[1], float(sg[1]), img.dtype, img.device)
    kernel = torch.mm(ky[:, None], kx[None, :])
    c = img.shape[-3]
    kernel = kernel.expand(c, 1, kernel.shape[0], kernel.shape[1])
    pad = [ks[0] // 2, ks[0] // 2, ks[1] // 2, ks[1] // 2]
    x = F.pad(img, pad, mode="reflect")
    return F.conv2d(x, kernel, groups=c)
