"""``DRR``: the reference's user-facing module, rendering on the MI355X.

API-compatible restatement of reference ``diffdrr/drr.py:23-266`` (constructor,
``forward``, ``render``, ``reshape_transform``, ``set_intrinsics_``,
``rescale_detector_``, the ``affine`` / ``n_patches`` / ``device`` / ``dtype``
properties, ``perspective_projection`` / ``inverse_projection``).  The only
behavioural difference is the renderer behind ``self.renderer``: the fused HIP
kernels of :mod:`diffdrr_amd.renderers` instead of the vectorised tensor
program, which makes ``patch_size`` and ``checkpoint_gradients`` unnecessary
(they are still honoured so that existing code runs unchanged).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from . import ops
from .detector import Detector
from .pose import RigidTransform, convert, euler_world_pose
from .renderers import Siddon, Trilinear


class DRR(nn.Module):
    """Differentiable digitally reconstructed radiographs from a CT subject."""

    def __init__(
        self,
        subject,  # object with .volume.affine, .density.data, .mask, .reorient
        sdd: float,  # source-to-detector distance
        height: int,
        delx: float,  # pixel size along x
        width: int | None = None,
        dely: float | None = None,
        x0: float = 0.0,  # principal point offsets
        y0: float = 0.0,
        p_subsample: float | None = None,  # proportion of pixels to render
        reshape: bool = True,  # return (B, C, H, W)
        reverse_x_axis: bool = True,  # radiologic convention
        patch_size: int | None = None,  # render the detector in sequential patches
        renderer: str = "siddon",
        voxel_shift: float = 0.5,
        persistent: bool = True,
        compile_renderer: bool = False,
        checkpoint_gradients: bool = False,
        **renderer_kwargs,
    ):
        super().__init__()
        width = height if width is None else width
        dely = delx if dely is None else dely
        n_subsample = int(height * width * p_subsample) if p_subsample is not None else None
        self.detector = Detector(sdd, height, width, delx, dely, x0, y0, subject.reorient,
                                 reverse_x_axis=reverse_x_axis, n_subsample=n_subsample)

        self.subject = subject
        affine = torch.as_tensor(subject.volume.affine, dtype=torch.float32).unsqueeze(0)
        self.register_buffer("_affine", affine, persistent=persistent)
        # (row-major: torch.linalg.inv hands back column-major strides, and every render would
        # copy the 3 x 4 block it passes to the kernels)
        self.register_buffer("_affine_inverse", torch.linalg.inv(affine).contiguous(),
                             persistent=persistent)
        density = subject.density.data.squeeze().to(torch.float32).contiguous()
        self.register_buffer("density", density, persistent=persistent)
        if subject.mask is not None:
            self.register_buffer("mask", subject.mask.data.to(torch.float32).squeeze(),
                                 persistent=persistent)

        if renderer == "siddon":
            self.renderer = Siddon(voxel_shift, **renderer_kwargs)
        elif renderer == "trilinear":
            self.renderer = Trilinear(voxel_shift, **renderer_kwargs)
        else:
            raise ValueError(f"renderer must be 'siddon' or 'trilinear', not {renderer}")
        # compile_renderer asked torch.compile to fuse the reference's tensor program;
        # the renderer here already is one hand-written kernel, so the flag is a no-op.
        self.compile_renderer = compile_renderer
        self.reshape = reshape
        self.patch_size = patch_size
        self.checkpoint_gradients = checkpoint_gradients

    # ------------------------------------------------------------ properties
    @property
    def affine(self):
        return RigidTransform(self._affine)

    @property
    def affine_inverse(self):
        return RigidTransform(self._affine_inverse)

    @property
    def n_patches(self):
        return (self.detector.height * self.detector.width) // (self.patch_size**2)

    @property
    def device(self):
        return self.density.device

    @property
    def dtype(self):
        return self.density.dtype

    def reshape_transform(self, img, batch_size):
        if self.reshape:
            if self.detector.n_subsample is None:
                img = img.view(batch_size, -1, self.detector.height, self.detector.width)
            else:
                img = reshape_subsampled_drr(img, self.detector, batch_size)
        return img

    def volume_changed(self):
        """``self.density`` was edited through ``.data`` (PyTorch does not track that): rebuild the
        renderer's cached bricks on the next render.  See ``Siddon.volume_changed`` -- without the
        call such an edit is still rendered from the live values, at the fp32 bricks' speed."""
        ops.invalidate_brick_workspace(self.density)

    # --------------------------------------------------------------- forward
    def forward(self, *args, parameterization: str = None, convention: str = None,
                calibration: RigidTransform = None, mask_to_channels: bool = False,
                degrees: bool = False, **kwargs):
        """Render DRRs for a batch of poses (``RigidTransform`` or raw parameters)."""
        fused = self._fused_ok(mask_to_channels, kwargs, calibration)
        if (fused and parameterization == "euler_angles" and len(args) == 2
                and all(torch.is_tensor(a) and a.dim() == 2 and a.shape[-1] == 3
                        and a.dtype == torch.float32 and ops.on_device(a) for a in args)):
            if (not mask_to_channels and calibration is None and not kwargs
                    and not (torch.is_grad_enabled() and (args[0].requires_grad or args[1].requires_grad
                                                          or self.density.requires_grad))):
                img = self._render_euler_inference(args[0], args[1], convention, degrees)
                if img is not None:
                    return self._reshape_fused(img, len(args[0]))
            if not mask_to_channels and calibration is None and not kwargs:
                img = self._render_euler_differentiable(args[0], args[1], convention, degrees)
                if img is not None:
                    return self._reshape_fused(img, len(args[0]))
            # pose parameters -> world matrix in one kernel (pose.py euler_world_pose)
            Mw = euler_world_pose(args[0], args[1], convention, self.detector._reorient,
                                  degrees=degrees)
            return self._reshape_fused(self._render_fused_Mw(Mw, calibration, mask_to_channels, **kwargs),
                                       len(Mw))
        if parameterization is None:
            pose = args[0]
        else:
            pose = convert(*args, parameterization=parameterization, convention=convention,
                           degrees=degrees)
        if fused:
            return self._reshape_fused(self._render_fused(pose, calibration, mask_to_channels, **kwargs),
                                       len(pose))
        source, target = self.detector(pose, calibration)
        # (rays straight out of the Detector: a row-major affine grid by construction)
        self._rays_from_detector = True
        try:
            if self.checkpoint_gradients:
                img = checkpoint(self.render, self.density, source, target, mask_to_channels,
                                 **kwargs, use_reentrant=False)
            else:
                img = self.render(self.density, source, target, mask_to_channels, **kwargs)
        finally:
            self._rays_from_detector = False
        return self.reshape_transform(img, batch_size=len(pose))

    def _reshape_fused(self, img, batch_size):
        """``reshape_transform`` of a fused render.  A subsample rendered through the brick kernels'
        pixel mask arrives as ``_ScatteredGrid``: the whole grid with zeros at the pixels that were
        not drawn -- which IS what ``reshape_subsampled_drr`` builds (reference drr.py:142-147) --
        so ``reshape=True`` is a view and ``reshape=False`` one gather."""
        if isinstance(img, _ScatteredGrid):
            det = self.detector
            if self.reshape:
                return img.dense.view(batch_size, -1, det.height, det.width)
            return img.dense.index_select(-1, det.subsample_index())
        return self.reshape_transform(img, batch_size=batch_size)

    # The DRR case end to end on the GPU: pose -> rays -> line integrals without the
    # (B, N, 3) ray tensors (and their gradients) passing through PyTorch ops.  Same maths
    # in the same order as `detector(...)` + `render(...)` (reference detector.py:144-154,
    # drr.py:191-227); used whenever nothing asks for a feature only the general path has.
    fuse_ray_generation = True
    _rays_from_detector = False  # set around the render call of forward()

    def _fused_ok(self, mask_to_channels, kwargs, cal=None, dense_only=False):
        """Whether the call can take the fused pose -> rays -> brick-kernel entries.  ``patch_size`` and
        ``p_subsample`` -- the reference's own speed levers (drr.py:36-39, 142-147, 218-225) -- stay on
        them (``_render_sparse``); ``dense_only``: callers that read the record of the whole grid
        (``ncc``) do not take a subsample."""
        r = self.renderer
        if dense_only and self.detector.n_subsample is not None:
            return False
        if cal is not None and getattr(getattr(cal, "matrix", None), "requires_grad", False):
            return False  # gradients w.r.t. the intrinsics flow through Detector.forward only
        # (the marcher's fused entry takes its one everyday keyword, n_points)
        kw_ok = not kwargs or (isinstance(r, Trilinear)
                               and set(kwargs) <= {"n_points", "alphamin", "alphamax"})
        return (self.fuse_ray_generation and isinstance(r, (Siddon, Trilinear))
                and r.supports_pose_entry()
                and ops.on_device(self.density) and self.density.dtype == torch.float32
                and (not mask_to_channels or getattr(self, "mask", None) is not None)
                and kw_ok and not self.checkpoint_gradients
                and (self.patch_size is None or self.n_patches >= 1)
                and min(self.detector.height, self.detector.width) >= 2)

    def _render_fused(self, pose, calibration, mask_to_channels=False, **kwargs):
        Mw = (pose.matrix @ self.detector._reorient)[:, :3, :]   # reorient.compose(extrinsic)
        return self._render_fused_Mw(Mw, calibration, mask_to_channels, **kwargs)

    def _calibrated_points(self, calibration=None):
        """(H W, 3) calibrated points of the WHOLE detector grid (detector.py:147-150; with
        ``p_subsample`` the grid the subsample was drawn from).  They only change with the
        intrinsics: cached per (calibration buffer version, grid buffer, device) -- the key holds
        the buffer objects themselves: alive, so not confusable with new ones."""
        det = self.detector
        grid = det.full_target()
        if calibration is not None:
            return calibration(grid)[0].detach()
        key = getattr(self, "_P_key", None)
        if key is None or key[0] is not det._calibration or key[1] != det._calibration._version \
                or key[2] is not grid or key[3] != grid._version:
            self._P_cache = det.calibration(grid)[0].detach()
            self._P_key = (det._calibration, det._calibration._version, grid, grid._version)
        return self._P_cache

    def _render_fused_Mw(self, Mw, calibration, mask_to_channels=False, **kwargs):
        det = self.detector
        P = self._calibrated_points(calibration)
        Ainv = self._affine_inverse[0, :3, :] if self._affine_inverse.dim() == 3 \
            else self._affine_inverse[:3, :]
        if self.patch_size is not None or det.n_subsample is not None:
            return self._render_sparse(Mw, P, Ainv, self.mask if mask_to_channels else None, **kwargs)
        # the grid contract holds for THIS call only (rays generated right here): a later direct
        # `drr.renderer(...)` call with rays of its own is checked again (renderers._grid_or_none)
        self.renderer.detector_shape = (det.height, det.width)
        self.renderer.trust_detector_shape = True
        try:
            return self.renderer.render_poses(self.density, Mw, P, Ainv,
                                              mask=self.mask if mask_to_channels else None,
                                              **kwargs)
        finally:
            self.renderer.trust_detector_shape = False

    def _sparse_plan(self):
        """``patch_size`` / ``p_subsample`` as the brick kernels see them: the rendered rays, in the
        renderer's output order (the subsample's order, else row-major), cut into the reference's
        chunks (``target.chunk(n_patches, dim=1)``, drr.py:218-225) -> per chunk (first detector
        row, number of rows, index of the chunk's rays inside those rows or None = all of them in
        order, the chunk's pixel indices in the whole grid or None = a contiguous run [a, b)).
        A chunk is rendered as the smallest run of whole detector rows that holds it -- itself a
        row-major affine grid, which is what the volume-stationary kernels take -- and gathered."""
        det = self.detector
        H, W = det.height, det.width
        # (cached ON the detector: a new detector -- set_intrinsics_ -- starts without a plan, whatever
        # address it is given)
        key = (H, W, self.patch_size, None if det.n_subsample is None else len(det.subsamples))
        plan = getattr(det, "_sparse_plan_cache", None)
        if plan is not None and plan[0] == key and plan[2] == det.target.device:
            return plan[1]
        idx = det.subsample_index()
        n = H * W if idx is None else int(idx.numel())
        if self.patch_size is not None:
            k = self.n_patches
            size = -(-n // k)  # torch.chunk: chunks of ceil(n / k) rays, the last one shorter
            bounds = [(a, min(a + size, n)) for a in range(0, n, size)]
        else:
            bounds = [(0, n)]
        host = None if idx is None else idx.cpu()
        chunks = []
        for a, b in bounds:
            if host is None:
                r0, r1 = a // W, (b - 1) // W
            else:
                r0, r1 = int(host[a:b].min()) // W, int(host[a:b].max()) // W
            if r1 == r0:  # (a grid has at least two rows)
                r0, r1 = (r0, r0 + 1) if r0 + 1 < H else (r0 - 1, r0)
            whole = host is None and a == r0 * W and b == (r1 + 1) * W
            local = None if whole else (
                torch.arange(a - r0 * W, b - r0 * W, device=det.target.device) if host is None
                else idx[a:b] - r0 * W)
            chunks.append((r0, r1 - r0 + 1, local, None if host is None else idx[a:b], (a, b)))
        det._sparse_plan_cache = (key, chunks, det.target.device)
        return chunks

    def _render_sparse(self, Mw, P, Ainv, mask, **kwargs):
        """``patch_size`` and / or ``p_subsample`` on the volume-stationary kernels.

        Siddon is per-ray independent: neither the chunking nor the choice of rays changes a ray's
        value, so the whole grid is rendered by ONE fused launch and the rendered rays
        are gathered from it -- patches cost nothing, a subsample costs what the grid costs (which
        is 3-12x less than the same rays through the per-ray kernels, profiles/r06/sparse.txt).
        The marcher's sample positions depend on the marching range of the rays of ONE renderer
        call (renderers.py:220-223): the range is taken over exactly the rays of each chunk, as
        the reference's patch loop does, and every chunk is rendered on the bricks as the run of
        whole detector rows that holds it (``_sparse_plan``), with that range."""
        from .renderers import _RaygenFn, get_alpha_minmax

        det, r = self.detector, self.renderer
        H, W = det.height, det.width
        idx = det.subsample_index()
        if isinstance(r, Siddon):
            # a subsample, plain render: the brick kernels drop the pixels that were not drawn right
            # after the candidate test (ddrr_siddon_forward_bricks_masked) -- a tenth of the walks at
            # p_subsample = 0.1, and the image they leave is the scattered one
            pm = det.subsample_mask() if (idx is not None and mask is None and r.grid_path == "bricks") else None
            r.detector_shape, r.trust_detector_shape = (H, W), True
            try:
                dense = r.render_poses(self.density, Mw, P, Ainv, mask=mask, pixel_mask=pm)  # (B, C, H W)
            finally:
                r.trust_detector_shape = False
            if pm is not None:
                return _ScatteredGrid(dense)
            return dense if idx is None else dense.index_select(-1, idx)
        source, target, img = _RaygenFn.apply(Mw, P, Ainv)
        given = kwargs.get("alphamin") is not None and kwargs.get("alphamax") is not None
        need_grad = torch.is_grad_enabled() and (source.requires_grad or target.requires_grad)
        outs = []
        for r0, rows, local, pix, (a, b) in self._sparse_plan():
            kw = dict(kwargs)
            t_rows = target[:, r0 * W:(r0 + rows) * W]
            i_rows = img[:, r0 * W:(r0 + rows) * W]
            if not given:
                # the reference's range of THIS renderer call: the chunk's own rays
                t_c = t_rows if local is None else t_rows.index_select(1, local)
                if need_grad:
                    lo, hi = get_alpha_minmax(source, t_c, r.dims(self.density), r.voxel_shift, r.eps)
                    kw["alphamin"], kw["alphamax"] = lo.min(), hi.max()
                else:
                    kw["alphamin"], kw["alphamax"] = ops.trilinear_alpha_range(
                        source.detach(), t_c.detach().contiguous(), self.density.shape,
                        voxel_shift=r.voxel_shift, eps=r.eps)
            r.detector_shape, r.trust_detector_shape = (rows, W), True
            try:
                out = r(self.density, source, t_rows.contiguous(), i_rows.contiguous(), mask=mask, **kw)
            finally:
                r.trust_detector_shape = False
            outs.append(out if local is None else out.index_select(-1, local))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)

    def _render_euler_inference(self, rot, xyz, convention, degrees):
        """The everyday call -- ``drr(rot, xyz, parameterization="euler_angles")`` with nothing to
        differentiate -- in TWO launches: pose -> matrix -> rays (which also clears the image and
        the brick counter of the render behind it), and the brick kernel.  The same kernels'
        arithmetic as the differentiable path (four launches: pose, rays, clear, render); None
        where that path does not apply (then the caller takes the usual one)."""
        from .pose import _AXIS, _check_convention
        from .renderers import _brick_storage

        r, det = self.renderer, self.detector
        B = rot.shape[0]
        if not (isinstance(r, Siddon) and r.grid_path == "bricks" and not r.packed_record and B > 0
                and rot.shape == xyz.shape and rot.device == self.density.device == xyz.device):
            return None
        _check_convention(convention)
        if degrees:
            rot = rot / 180 * math.pi
        axes = tuple(_AXIS[c] for c in convention)
        P = self._calibrated_points()
        Ainv = self._affine_inverse[0, :3, :] if self._affine_inverse.dim() == 3 \
            else self._affine_inverse[:3, :]
        cfg = r._cfg(False, det=(det.height, det.width))
        with torch.no_grad():
            out = torch.empty(B, P.shape[0], dtype=torch.float32, device=rot.device)
            launch_ws = ops.launch_workspace(self.density.shape, self.density.device)
            _, source, target, img = ops.pose_raygen_forward(
                rot.detach(), xyz.detach(), axes, det._reorient[:3, :].contiguous(), Ainv, P, clear=out,
                clear_launch_ws=launch_ws)
            # (the density tensor itself: its packed bricks are cached per tensor object and version)
            ops.siddon_forward_bricks(self.density, source, target, img, cfg["det"],
                                      voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
                                      storage=_brick_storage(self.density, cfg, B), out=out, launch_ws=launch_ws,
                                      cleared=True, pixel_mask=det.subsample_mask())
        if det.n_subsample is not None:  # (p_subsample: the scattered grid, see _reshape_fused)
            return _ScatteredGrid(out.unsqueeze(1))
        return out.unsqueeze(1)

    def _render_euler_differentiable(self, rot, xyz, convention, degrees):
        """``drr(rot, xyz, parameterization="euler_angles")`` with pose parameters that take a gradient
        and a similarity computed outside (any criterion of ``metrics``, a user's loss): the render's
        forward in three launches and its backward in one (``renderers._EulerSiddonImageFn``) where the
        composition of ``euler_world_pose`` and ``render_poses`` takes five and three.  None where that
        does not apply -- a dense Siddon render on the bricks of a volume that takes no gradient, at most
        ``FUSED_NCC_MAX_POSES`` poses (beyond, the launches are bound by their bytes, not their count)."""
        from .pose import _AXIS, _check_convention
        from .renderers import _EulerSiddonImageFn

        r, det = self.renderer, self.detector
        B = rot.shape[0]
        if not (torch.is_grad_enabled() and (rot.requires_grad or xyz.requires_grad)
                and not self.density.requires_grad and isinstance(r, Siddon) and r.grid_path == "bricks"
                and not r.packed_record and 0 < B <= self.FUSED_NCC_MAX_POSES and rot.shape == xyz.shape
                and rot.device == self.density.device == xyz.device
                and self.patch_size is None and det.n_subsample is None and self.fuse_ray_generation):
            return None
        _check_convention(convention)
        if degrees:
            rot = rot / 180 * math.pi
        axes = tuple(_AXIS[c] for c in convention)
        P = self._calibrated_points()
        Ainv = self._affine_inverse[0, :3, :] if self._affine_inverse.dim() == 3 \
            else self._affine_inverse[:3, :]
        cfg = r._cfg(False, det=(det.height, det.width))
        out = _EulerSiddonImageFn.apply(rot, xyz, self.density, det._reorient[:3, :].contiguous(), P, Ainv,
                                        axes, cfg)
        return out.unsqueeze(1)

    FUSED_NCC_MAX_POSES = 32

    def ncc(self, fixed: torch.Tensor, rot: torch.Tensor, xyz: torch.Tensor, *,
            convention: str = "ZXY", degrees: bool = False, eps: float = 1e-5,
            reduction: str = "none") -> torch.Tensor:
        """Per-pose normalised cross-correlation of ``fixed`` ((1 | B), 1, H, W) with the DRRs at the
        Euler poses (rot, xyz) (B, 3): ``NormalizedCrossCorrelation2d(eps=eps)(fixed.expand(B, ...),
        self(rot, xyz, parameterization="euler_angles", convention=convention))`` -- the objective of
        the reference's registration loop (registration.py:32-42 + metrics.py:21-44) -- -> (B,).

        When the pose parameters require a gradient and the render takes the brick kernel, the
        whole step runs as three fused launches around it instead of nine (pose -> matrix -> rays;
        image from the backward record + NCC; NCC backward -> ray gradients -> matrix -> pose
        parameters: ``renderers._EulerSiddonNccFn``): 13 % of a one-pose registration iteration
        (0.246 -> 0.213 ms; up to ``FUSED_NCC_MAX_POSES`` poses per call).
        Anything else (no gradient wanted: the forward-only kernel is the faster one; other
        renderers, subsampling, a volume that requires a gradient) composes the same
        result from ``forward`` and the NCC module.

        ``reduction="sum"``: the batch's objective ``sum_b ncc_b`` as a 0-dim tensor -- what a batched
        registration step maximises.  On the fused route the sum comes out of the epilogue's own
        launch and its gradient goes back in as one value: ``loss.backward(gradient=one)`` with a
        ready-made ``one`` is a step without a reduction or a fill launch.  The per-pose values of
        the call (detached, (B,)) are left in ``self.ncc_per_pose``."""
        from .metrics import NormalizedCrossCorrelation2d
        from .pose import _AXIS, _check_convention
        from .renderers import _EulerSiddonNccFn

        B = rot.shape[0]
        r, det = self.renderer, self.detector
        # (at most FUSED_NCC_MAX_POSES poses: the launches around the brick kernel are bound by their
        # bytes, not by their count, from a few dozen poses on -- 16 poses 0.821 against 0.875 ms per
        # eager step; 32 poses 1.481 against 1.488 at 512^3, 0.721 against 0.751 at 256^3 since the
        # epilogues' last workgroups read their sums in one round trip (1.480 against 1.463 before);
        # not measured beyond: profiles/r05/fused_step.txt)
        ok = (torch.is_grad_enabled() and (rot.requires_grad or xyz.requires_grad)
              and B <= self.FUSED_NCC_MAX_POSES
              and self._fused_ok(False, {}, None, dense_only=True) and isinstance(r, Siddon)
              and r.grid_path == "bricks" and not r.packed_record and not self.density.requires_grad
              and all(torch.is_tensor(a) and a.dim() == 2 and a.shape == (B, 3)
                      and a.dtype == torch.float32 and ops.on_device(a) for a in (rot, xyz))
              and torch.is_tensor(fixed) and fixed.dim() == 4 and fixed.shape[0] in (1, B)
              and fixed.shape[1:] == (1, det.height, det.width) and fixed.dtype == torch.float32
              and fixed.device == rot.device and not fixed.requires_grad and B > 0)
        if reduction not in ("none", "sum"):
            raise ValueError(f"reduction must be 'none' or 'sum', not {reduction}")
        if not ok:
            img = self(rot, xyz, parameterization="euler_angles", convention=convention, degrees=degrees)
            if img.dim() == 3 and det.n_subsample is None:
                # DRR(reshape=False) hands (B, C, N) back: the criterion wants the detector's grid,
                # as the fused path reads it
                img = img.view(B, -1, det.height, det.width)
            vals = NormalizedCrossCorrelation2d(eps=eps)(fixed.expand(B, -1, -1, -1), img)
            if reduction == "sum":
                self.ncc_per_pose = vals.detach()
                return vals.sum()
            return vals
        _check_convention(convention)
        if degrees:
            rot = rot / 180 * math.pi
        axes = tuple(_AXIS[c] for c in convention)
        # (as _render_fused_Mw: the calibrated detector points, cached per intrinsics)
        self._calibrated_points()
        Ainv = self._affine_inverse[0, :3, :] if self._affine_inverse.dim() == 3 \
            else self._affine_inverse[:3, :]
        cfg = r._cfg(False, det=(det.height, det.width))
        x1 = fixed.reshape(fixed.shape[0], det.height * det.width)
        res = _EulerSiddonNccFn.apply(rot, xyz, self.density, det._reorient[:3, :].contiguous(),
                                      self._P_cache, Ainv, x1, axes, cfg, float(eps), reduction == "sum")
        if reduction == "sum":
            total, self.ncc_per_pose = res
            return total
        return res

    @torch.no_grad()
    def marching_range(self, *args, parameterization: str = None, convention: str = None,
                       calibration: RigidTransform = None, degrees: bool = False):
        """(alphamin, alphamax), 0-dim tensors: the marcher's batch-global range (reference
        renderers.py:220-223) of the rays of these poses, without rendering them -- what a
        caller passes back as ``alphamin=`` / ``alphamax=`` to render several batches on one
        common range (``diffdrr_amd.dist.sweep`` does, so that a sharded sweep does not depend
        on the world size)."""
        pose = args[0] if parameterization is None else convert(
            *args, parameterization=parameterization, convention=convention, degrees=degrees)
        source, target = self.detector(pose, calibration)
        source, target = self.affine_inverse(source), self.affine_inverse(target)
        r = self.renderer
        if ops.on_device(target) and target.dtype == torch.float32:
            return ops.trilinear_alpha_range(source, target, self.density.shape,
                                             voxel_shift=r.voxel_shift, eps=r.eps)
        from .renderers import get_alpha_minmax
        lo, hi = get_alpha_minmax(source, target, r.dims(self.density), r.voxel_shift, r.eps)
        return lo.min(), hi.max()

    def render(self, density: torch.Tensor, source: torch.Tensor, target: torch.Tensor,
               mask_to_channels: bool = False, **kwargs):
        """World-space rays -> ``(B, C, N)`` line integrals (reference drr.py:191-227)."""
        # ray length in world units, before the rays go to voxel space
        img = (target - source).norm(dim=-1).unsqueeze(1)
        source = self.affine_inverse(source)
        target = self.affine_inverse(target)

        kwargs["mask"] = self.mask if mask_to_channels else None
        r = self.renderer
        # Siddon is per-ray independent: where the whole grid goes to the volume-stationary kernel
        # (nothing is materialised per segment there), the patch loop changes nothing but the
        # number of launches -- one render (the reference's chunking exists to bound the memory of
        # its (B, N, M) tensors, drr.py:218-225)
        unpatched = self.patch_size is None or (
            isinstance(r, Siddon) and r.supports_pose_entry() and r.grid_path == "bricks"
            and ops.on_device(density) and density.dtype == torch.float32
            and not kwargs.get("align_corners", False) and self.n_patches >= 1)
        full_grid = (self.detector.n_subsample is None and unpatched and
                     target.shape[1] == self.detector.height * self.detector.width)
        # `detector_shape` is a CONTRACT with the renderer, not a hint: the volume-stationary
        # kernels cull rays with an affine model of the detector grid.  Rays that forward() got
        # from the Detector satisfy it by construction; rays handed to render() directly (the
        # tutorials do, reconstruction.ipynb:122) are checked once (one reduction, one sync) and
        # rendered by the per-ray kernels if they are not such a grid (permuted, subsampled ...).
        if full_grid and not self._rays_from_detector and ops.on_device(target):
            full_grid = ops.rays_form_detector_grid(source, target, self.detector.height,
                                                    self.detector.width)
        self.renderer.detector_shape = \
            (self.detector.height, self.detector.width) if full_grid else None
        self.renderer.trust_detector_shape = True  # generated or checked right here ...
        try:
            if unpatched and (self.patch_size is None or full_grid):
                return self.renderer(density, source, target, img, **kwargs)
            partials = [
                self.renderer(density, source, t, i, **kwargs)
                for t, i in zip(target.chunk(self.n_patches, dim=1),
                                img.chunk(self.n_patches, dim=-1))
            ]
            return torch.cat(partials, dim=-1)
        finally:
            # ... and for this call only: rays handed to `drr.renderer(...)` directly afterwards
            # (the trilinear tutorial does) are verified again before the brick kernels see them
            self.renderer.trust_detector_shape = False

    # ------------------------------------------------------------ intrinsics
    def set_intrinsics_(self, sdd: float = None, height: int = None, width: int = None,
                        delx: float = None, dely: float = None, x0: float = None,
                        y0: float = None, n_subsample: int = None, reverse_x_axis: bool = None):
        """Replace the detector in place (reference drr.py:230-255)."""
        d = self.detector
        self.detector = Detector(
            sdd if sdd is not None else d.sdd,
            height if height is not None else d.height,
            width if width is not None else d.width,
            delx if delx is not None else d.delx,
            dely if dely is not None else d.dely,
            x0 if x0 is not None else -d.x0,
            y0 if y0 is not None else -d.y0,
            self.subject.reorient,
            n_subsample if n_subsample is not None else d.n_subsample,
            reverse_x_axis if reverse_x_axis is not None else d.reverse_x_axis,
        ).to(self.density)

    def rescale_detector_(self, scale: float):
        """Rescale the detector plane in place (reference drr.py:258-266)."""
        self.set_intrinsics_(
            height=int(self.detector.height * scale),
            width=int(self.detector.width * scale),
            delx=float(self.detector.delx / scale),
            dely=float(self.detector.dely / scale),
        )

    # ------------------------------------------------------------ projections
    def perspective_projection(self, pose: RigidTransform, pts: torch.Tensor):
        """World points (3D) -> pixel coordinates (2D) (reference drr.py:269-291)."""
        extrinsic = (self.detector.reorient.compose(pose)).inverse()
        x = extrinsic(pts)
        x = torch.einsum("ij,bnj->bni", self.detector.intrinsic, x)
        x = x / x[..., -1:].clone()
        u = self.detector.width - x[..., 0] if self.detector.reverse_x_axis else x[..., 0]
        v = self.detector.height - x[..., 1]
        return torch.stack([u, v], dim=-1)

    def inverse_projection(self, pose: RigidTransform, pts: torch.Tensor):
        """Pixel coordinates (2D) -> points on the detector plane in world space."""
        v = self.detector.height - pts[..., 1]
        u = self.detector.width - pts[..., 0] if self.detector.reverse_x_axis else pts[..., 0]
        uv1 = torch.stack([u, v, torch.ones_like(u)], dim=-1)
        x = self.detector.sdd * torch.einsum(
            "ij,bnj->bni", torch.linalg.inv(self.detector.intrinsic), uv1)
        return self.detector.reorient.compose(pose)(x)


class _ScatteredGrid:
    """A subsample as the masked brick kernels leave it: ``dense`` (B, 1, H W), zeros at the pixels
    that were not drawn (``DRR._reshape_fused``)."""
    __slots__ = ("dense",)

    def __init__(self, dense):
        self.dense = dense


def reshape_subsampled_drr(img: torch.Tensor, detector: Detector, batch_size: int):
    """(B, 1, n) subsampled line integrals -> (B, 1, H, W) with zeros elsewhere (reference
    drr.py:142-147).  The indices come from the detector's cached device tensor: indexing with the
    Python list ``subsamples[-1]``, as the reference does, builds and uploads a 4000-element
    tensor per call -- 6 ms of a 0.1 ms render (profiles/r06/sparse.txt)."""
    n_points = detector.height * detector.width
    idx = detector.subsample_index()
    if idx.device != img.device:
        idx = idx.to(img.device)
    # (the reference's `drr[:, idx] = img` only broadcasts for batch_size == 1)
    drr = torch.zeros(batch_size, n_points, dtype=img.dtype, device=img.device)
    drr = drr.index_copy(1, idx, img.reshape(batch_size, -1))
    return drr.view(batch_size, 1, detector.height, detector.width)
