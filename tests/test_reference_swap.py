"""INTEGRATION.md section 2, exercised: an UNMODIFIED reference ``diffdrr.drr.DRR`` object
re-pointed at ``diffdrr_amd.Siddon`` / ``diffdrr_amd.Trilinear`` (``drr.renderer = ...``) renders
and back-propagates like the reference's own renderer.  Runs in the build container only
(the reference lives at /root/reference; on the GPU box this module skips): the kernels are
the host emulation of the same cores, the Python / autograd layer is the product's."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference checkout not present")


def _reference_drr(ref, renderer, D=32, det=20):
    vol = torch.rand(D, D, D, generator=torch.Generator().manual_seed(0))
    affine = np.diag([1.5, 1.5, 1.5, 1.0])
    affine[:3, 3] = -(D - 1) / 2 * 1.5
    subject = ref.Subject(volume=ref.ScalarImage(vol[None], affine),
                          density=ref.ScalarImage(vol[None], affine),
                          reorient=torch.tensor([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]]),
                          mask=None, fiducials=None)
    return ref.DRR(subject, sdd=400.0, height=det, delx=3.0, renderer=renderer)


@pytest.mark.parametrize("renderer", ["siddon", "trilinear"])
def test_swapped_renderer_in_the_reference_drr(emulated_ops, renderer):
    import diffdrr_amd

    ref = ref_loader.load()
    rot0 = torch.tensor([[0.3, -0.2, 0.25], [0.0, 0.4, -0.1]])
    xyz0 = torch.tensor([[4.0, 260.0, -3.0], [-2.0, 250.0, 5.0]])
    kw = {} if renderer == "siddon" else {"n_points": 60}
    res = {}
    for swapped in (False, True):
        drr = _reference_drr(ref, renderer)
        if swapped:
            drr.renderer = (diffdrr_amd.Siddon if renderer == "siddon" else diffdrr_amd.Trilinear)(
                voxel_shift=0.5)
            if renderer == "siddon":
                # the contract of INTEGRATION.md: the rays DRR.forward hands over are the
                # row-major detector grid -> the volume-stationary kernels may take them
                drr.renderer.detector_shape = (drr.detector.height, drr.detector.width)
        rot = rot0.clone().requires_grad_()
        xyz = xyz0.clone().requires_grad_()
        img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", **kw)
        W = torch.rand(img.shape, generator=torch.Generator().manual_seed(3))
        (img * W).sum().backward()
        res[swapped] = (img.detach().numpy(), rot.grad.numpy(), xyz.grad.numpy())
    assert res[True][0].shape == res[False][0].shape
    assert rel_err(res[True][0], res[False][0]) < 1e-4          # forward: the north-star tolerance
    # pose gradients through the reference's own Detector / pose code and our autograd Functions
    # (noise volume, nearest voxels: the reference's fp32 gradient is itself ~1e-2 from fp64)
    tol = 5e-2 if renderer == "siddon" else 5e-3
    assert rel_err(res[True][1], res[False][1]) < tol
    assert rel_err(res[True][2], res[False][2]) < tol
