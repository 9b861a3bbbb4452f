"""Subjects for the renderer: the conventions of reference ``diffdrr/data.py``
without its file I/O (torchio / nibabel are out of scope).

A *subject* is any object with ``.volume.affine`` (4x4 voxel->world),
``.density.data`` ((1,)Dx,Dy,Dz in [0, 1]), ``.mask`` (``None`` or ``.data``
label map) and ``.reorient`` (4x4): exactly what ``DRR.__init__`` reads
(reference drr.py:64-89).  ``Subject`` below is a minimal such object and
``synthetic_subject`` builds the seeded scenes of SURVEY.md section 8(d).
"""
from __future__ import annotations

import numpy as np
import torch

# Frame-of-reference changes (reference data.py:87-120)
REORIENT = {
    "AP": [[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]],
    "PA": [[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]],
    None: [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
}


class Image:
    def __init__(self, data: torch.Tensor, affine):
        self.data = data
        self.affine = np.asarray(affine, dtype=np.float64)


class Subject:
    def __init__(self, volume: Image, density: Image, reorient, mask: Image | None = None,
                 fiducials=None):
        self.volume = volume
        self.density = density
        self.mask = mask
        self.reorient = torch.as_tensor(reorient, dtype=torch.float32)
        self.fiducials = fiducials


def centered_affine(dims, spacing):
    """Voxel->world affine with the volume centre at the world origin, like
    reference ``canonicalize`` (data.py:187-202)."""
    A = np.diag([float(spacing[0]), float(spacing[1]), float(spacing[2]), 1.0])
    A[:3, 3] = [-(d - 1) / 2 * s for d, s in zip(dims, spacing)]
    return A


def reorient_matrix(orientation):
    if orientation not in REORIENT:
        raise ValueError(f"Unrecognized orientation {orientation}")
    return torch.tensor(REORIENT[orientation], dtype=torch.float32)


def transform_hu_to_density(volume: torch.Tensor, bone_attenuation_multiplier: float = 1.0):
    """HU -> [0, 1] density with air / soft tissue / bone classes (data.py:214-227)."""
    volume = volume.to(torch.float32)
    air = volume <= -800
    soft = (volume > -800) & (volume <= 350)
    bone = volume > 350
    density = torch.empty_like(volume)
    density[air] = volume[soft].min()
    density[soft] = volume[soft]
    density[bone] = volume[bone] * bone_attenuation_multiplier
    density -= density.min()
    density /= density.max()
    return density


def make_subject(volume: torch.Tensor, spacing=(1.0, 1.0, 1.0), orientation="AP", mask=None,
                 affine=None) -> Subject:
    """Wrap a (Dx,Dy,Dz) density array (and optional label map) as a subject."""
    A = centered_affine(volume.shape, spacing) if affine is None else affine
    vol = Image(volume.unsqueeze(0), A)
    m = None if mask is None else Image(mask.unsqueeze(0), A)
    return Subject(vol, Image(volume.unsqueeze(0), A), reorient_matrix(orientation), m)


def noise_volume(D, seed=0) -> torch.Tensor:
    """Volume A of SURVEY.md section 8(d): seeded uniform noise in [0, 1)."""
    dims = (D, D, D) if isinstance(D, int) else tuple(D)
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*dims, generator=g)


def phantom_volume(D, seed=0, n_blobs=24) -> torch.Tensor:
    """Volume B of SURVEY.md section 8(d): Gaussian ellipsoids + 1 % noise (smooth, so
    image similarity has a useful gradient for registration)."""
    dims = (D, D, D) if isinstance(D, int) else tuple(D)
    g = torch.Generator().manual_seed(seed)
    axes = [torch.linspace(-1, 1, d) for d in dims]
    vol = torch.zeros(dims)
    for _ in range(n_blobs):
        c = torch.rand(3, generator=g) * 1.2 - 0.6
        w = torch.rand(3, generator=g) * 0.25 + 0.08
        amp = torch.rand(1, generator=g).item() * 0.8 + 0.2
        gx = torch.exp(-(((axes[0] - c[0]) / w[0]) ** 2))
        gy = torch.exp(-(((axes[1] - c[1]) / w[1]) ** 2))
        gz = torch.exp(-(((axes[2] - c[2]) / w[2]) ** 2))
        vol += amp * gx[:, None, None] * gy[None, :, None] * gz[None, None, :]
    vol += 0.01 * torch.rand(dims, generator=g)
    vol -= vol.min()
    vol /= vol.max()
    return vol


def ct_like_hu_volume(dims=(512, 512, 133), seed=0) -> torch.Tensor:
    """A CT-like volume in Hounsfield units with the shape of the reference's example CT
    (512 x 512 x 133 at 0.703 x 0.703 x 2.5 mm; the CT itself is not shipped): air at -1000 HU
    around an elliptical body whose skin is a partial-volume ramp 1.5 voxels wide, soft tissue
    at 40 HU +- noise with smooth organ-scale variation, two lungs around -850 HU with texture
    (part of it above the -800 HU air threshold), a spine with a cortical shell, ribs as an
    interrupted thin bone shell, and one metal marker at 3000 HU.  Meant to go through
    :func:`transform_hu_to_density` (reference data.py:214-227): exact-zero air, dim lung voxels
    among zeros, partial-volume skin, bone near 0.5, the marker at 1.0 -- what the 16-bit brick
    storage's guard (csrc/brick_step.h q16_usable) has to cope with on a real scan."""
    Dx, Dy, Dz = dims
    g = torch.Generator().manual_seed(seed)
    x = (torch.arange(Dx, dtype=torch.float32) - (Dx - 1) / 2)[:, None, None]
    y = (torch.arange(Dy, dtype=torch.float32) - (Dy - 1) / 2)[None, :, None]
    z = (torch.arange(Dz, dtype=torch.float32) - (Dz - 1) / 2)[None, None, :]
    ax, ay = 0.42 * Dx, 0.30 * Dy                       # body: an elliptical cylinder along z
    rho = torch.sqrt((x / ax) ** 2 + (y / ay) ** 2)     # 1 on the skin
    # distance to the skin in voxels (first order), positive inside
    inside = (1.0 - rho) * min(ax, ay)
    body = (inside / 1.5 + 0.5).clamp(0.0, 1.0).expand(Dx, Dy, Dz)  # partial-volume ramp
    tissue = 40.0 + 15.0 * torch.randn(Dx, Dy, Dz, generator=g)
    for _ in range(6):                                  # organ-scale variation, +-20 HU
        c = (torch.rand(3, generator=g) - 0.5) * torch.tensor([0.6 * Dx, 0.4 * Dy, 0.8 * Dz])
        w = torch.tensor([0.12 * Dx, 0.10 * Dy, 0.25 * Dz]) * (0.6 + torch.rand(3, generator=g))
        amp = (torch.rand(1, generator=g).item() - 0.5) * 40.0
        tissue += amp * (torch.exp(-((x - c[0]) / w[0]) ** 2) * torch.exp(-((y - c[1]) / w[1]) ** 2)
                         * torch.exp(-((z - c[2]) / w[2]) ** 2))
    hu = -1000.0 + body * (tissue + 1000.0)
    for sx in (-1.0, 1.0):                              # lungs
        lung = (((x - sx * 0.20 * Dx) / (0.15 * Dx)) ** 2 + ((y + 0.02 * Dy) / (0.17 * Dy)) ** 2
                + (z / (0.42 * Dz)) ** 2) < 1.0
        tex = -850.0 + 60.0 * torch.randn(Dx, Dy, Dz, generator=g)
        hu = torch.where(lung, tex, hu)
    # spine: cancellous core (300 HU: soft-tissue class) in a cortical shell (1200 HU)
    rs = torch.sqrt((x / (0.045 * Dx)) ** 2 + ((y - 0.17 * Dy) / (0.045 * Dy)) ** 2).expand(Dx, Dy, Dz)
    hu = torch.where(rs < 1.0, torch.full_like(hu, 300.0), hu)
    hu = torch.where((rs >= 0.8) & (rs < 1.0), torch.full_like(hu, 1200.0), hu)
    # ribs: a shell 3 voxels thick just under the skin, 3 of every 8 slices
    rib = ((inside > 6.0) & (inside < 9.0)).expand(Dx, Dy, Dz) & ((torch.arange(Dz) % 8) < 3)[None, None, :]
    hu = torch.where(rib, 900.0 + 50.0 * torch.randn(Dx, Dy, Dz, generator=g), hu)
    # one metal marker
    mx, my, mz = Dx // 2 + Dx // 7, Dy // 2 - Dy // 9, Dz // 2 + 5
    hu[mx:mx + 3, my:my + 3, mz:mz + 2] = 3000.0
    return hu.contiguous()


def synthetic_subject(D, kind="noise", spacing=1.0, orientation="AP", seed=0,
                      n_labels=0) -> Subject:
    vol = noise_volume(D, seed) if kind == "noise" else phantom_volume(D, seed)
    sp = (spacing,) * 3 if np.isscalar(spacing) else tuple(spacing)
    mask = None
    if n_labels:
        g = torch.Generator().manual_seed(seed + 1)
        # piecewise-constant labels: coarse random blocks
        coarse = torch.randint(0, n_labels, tuple(max(1, d // 8) for d in vol.shape), generator=g)
        mask = coarse
        for ax, d in enumerate(vol.shape):
            idx = (torch.arange(d) * coarse.shape[ax] // d).clamp_max(coarse.shape[ax] - 1)
            mask = mask.index_select(ax, idx)
    return make_subject(vol, sp, orientation, mask)
