"""2D/3D registration module (reference ``diffdrr/registration.py:14-50``):
learnable pose parameters in front of a ``DRR``.  ``PoseRegressor`` (a CNN) is
out of scope."""
from __future__ import annotations

import torch
import torch.nn as nn

from .pose import convert


class Registration(nn.Module):
    def __init__(self, drr, rotation: torch.Tensor, translation: torch.Tensor,
                 parameterization: str, convention: str | None = None):
        super().__init__()
        self.drr = drr
        self._rotation = nn.Parameter(rotation)
        self._translation = nn.Parameter(translation)
        self.parameterization = parameterization
        self.convention = convention

    def forward(self, **kwargs):
        # (same result as the reference's `self.drr(self.pose, **kwargs)`; handing the raw
        # parameters over lets DRR take its fused pose -> rays path for Euler angles)
        return self.drr(self._rotation, self._translation,
                        parameterization=self.parameterization, convention=self.convention,
                        **kwargs)

    @property
    def pose(self):
        return convert(self._rotation, self._translation,
                       parameterization=self.parameterization, convention=self.convention)

    @property
    def rotation(self):
        return self._rotation

    @property
    def translation(self):
        return self._translation


class PoseAdam(torch.optim.Optimizer):
    """``torch.optim.Adam([{"params": [rotation], "lr": lr_rotation}, {"params": [translation],
    "lr": lr_translation}], betas=betas, eps=eps, maximize=maximize)`` -- the optimizer of the
    reference's registration loop (``notebooks/tutorials/registration.ipynb:240-316``) -- with the
    step of both groups in ONE launch (``ddrr_pose_adam_step``): torch's fused / capturable Adam
    takes four (a step-counter and an update launch per group), ~19 us of a 0.19 ms iteration at
    512^3 -> 256^2.  Same update rule (no weight decay, no amsgrad), same state layout
    (``state[p] = {"step", "exp_avg", "exp_avg_sq"}``, the counters on the device: safe to capture
    in a HIP graph), same param groups (learning rates may be changed between EAGER steps through
    ``param_groups``; they are host numbers handed to the kernel as arguments, so a captured step
    keeps the values it was captured with -- ``GraphedIteration`` raises rather than replay a stale
    learning rate).  Parameters: float32 ``(B, 3)`` on the GPU."""

    def __init__(self, rotation, translation, lr_rotation, lr_translation, betas=(0.9, 0.999), eps=1e-8,
                 maximize=False):
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps < 0.0:
            raise ValueError("PoseAdam: invalid betas / eps")
        if lr_rotation < 0.0 or lr_translation < 0.0:
            raise ValueError("PoseAdam: negative learning rate")
        super().__init__([{"params": [rotation], "lr": lr_rotation}, {"params": [translation], "lr": lr_translation}],
                         dict(lr=lr_rotation, betas=betas, eps=eps, maximize=maximize))
        # (the state exists from the start, not from the first step: a first step inside a graph
        # capture would otherwise bake the zero-fills of its state into every replay)
        for p in (rotation, translation):
            self._state(p)

    def _state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import ops

        g_rot, g_xyz = self.param_groups
        rot, xyz = g_rot["params"][0], g_xyz["params"][0]
        if rot.grad is None or xyz.grad is None:
            raise RuntimeError("PoseAdam.step(): both pose parameters need a gradient")
        if (g_rot["betas"], g_rot["eps"], g_rot["maximize"]) != (g_xyz["betas"], g_xyz["eps"], g_xyz["maximize"]):
            raise ValueError("PoseAdam: betas, eps and maximize are shared by the two groups")
        s_rot, s_xyz = self._state(rot), self._state(xyz)
        ops.pose_adam_step(rot, xyz, rot.grad.contiguous(), xyz.grad.contiguous(), s_rot["exp_avg"],
                           s_rot["exp_avg_sq"], s_xyz["exp_avg"], s_xyz["exp_avg_sq"], s_rot["step"],
                           s_xyz["step"], lr_rot=g_rot["lr"], lr_xyz=g_xyz["lr"], betas=g_rot["betas"],
                           eps=g_rot["eps"], maximize=g_rot["maximize"])
        return loss


class GraphedIteration:
    """One registration iteration -- render, similarity, backward, optimizer step -- captured
    once as a HIP graph and replayed: the loop of reference
    ``notebooks/tutorials/registration.ipynb:240-316`` without the ~0.5 ms of Python / autograd
    / launch overhead per iteration that dwarfs its ~0.25 ms of GPU work at 512^3 -> 256^2
    (``profiles/r01/config4_registration_v2.txt``).

        step = GraphedIteration(reg, criterion, optimizer, target)
        for it in range(n):
            loss = step()          # replays the graph; `loss` is a device tensor (no sync)

    The optimizer must not synchronise in ``step()`` (``torch.optim.SGD``; Adam with
    ``capturable=True``).  Everything the iteration touches keeps its address: the parameters
    of ``reg`` and the optimizer state are updated in place, ``target`` is read in place.
    ``maximize`` / learning rates are whatever the optimizer was built with -- and FROZEN at
    capture: hyper-parameters that are Python numbers (``PoseAdam``'s and torch's ``lr``, ``betas``,
    ``eps``) are baked into the captured kernel arguments, so an ``lr_scheduler`` or a manual
    ``param_groups`` edit between replays would be ignored silently.  ``__call__`` therefore raises
    if a group's host-side hyper-parameters differ from the captured ones (build a new
    ``GraphedIteration`` after changing them; a tensor ``lr`` is read by the kernels and may change).

    Construction has no side effects on the optimisation: the ``warmup`` eager iterations and
    the capture itself (which runs the iteration once more) are undone -- parameters and
    optimizer state are restored to what they were -- so that ``step()`` number k is iteration
    k of the loop, as in the reference's.  ``iterations_done`` counts the replays.
    Limits of that undo: optimizer state created by the warm-up is ZEROED, which equals a fresh
    start for Adam and for SGD with momentum and ``dampening == 0``; with ``dampening != 0`` a
    fresh first step sets ``buf = grad`` while a zeroed buffer yields ``(1 - dampening) grad``
    (rejected below), and state that is not a tensor (Python-int step counts of non-capturable
    optimizers) is not restored.

    ``static_volume`` (default True): the CT volume is not edited in place between replays --
    the graph then renders from the volume's cached 16-bit bricks (``Siddon.brick_storage``),
    whose address and "already built" state are baked into the graph.  Pass False if the volume
    changes under the graph: the captured render then reads the live fp32 volume.

    ``fused_similarity`` (default True): with ``NormalizedCrossCorrelation2d()`` as the criterion
    and Euler pose parameters the iteration goes through ``DRR.ncc`` -- three fused launches around
    the brick kernel instead of nine -- where that applies (``self.fused_similarity`` says whether
    it was asked for; the value and the gradients are those of the criterion on the rendered
    image, up to the order of the sums)."""

    def __init__(self, reg: Registration, criterion, optimizer, target: torch.Tensor,
                 warmup: int = 3, static_volume: bool = True, fused_similarity: bool = True,
                 **render_kwargs):
        self.reg, self.criterion, self.optimizer, self.target = reg, criterion, optimizer, target
        self.render_kwargs = render_kwargs
        self.iterations_done = 0
        for g in optimizer.param_groups:
            if g.get("momentum", 0) and g.get("dampening", 0):
                raise ValueError("GraphedIteration: SGD with momentum and dampening != 0 cannot "
                                 "be restored to a fresh first step after warm-up and capture")
        renderer = getattr(reg.drr, "renderer", None)
        had_static = getattr(renderer, "static_volume", None)
        if had_static is not None:
            renderer.static_volume = bool(static_volume)
        params = [p for g in optimizer.param_groups for p in g["params"]]
        saved_params = [p.detach().clone() for p in params]
        saved_state = {p: {k: v.detach().clone() for k, v in optimizer.state.get(p, {}).items()
                           if torch.is_tensor(v)} for p in params}

        # NCC of Euler poses: the step around the renderer as three fused launches (DRR.ncc; it
        # composes the same value from `reg()` and the criterion itself where they do not apply)
        from .metrics import NormalizedCrossCorrelation2d
        fused = (type(criterion) is NormalizedCrossCorrelation2d and criterion.patch_size is None
                 and reg.parameterization == "euler_angles" and not render_kwargs
                 and hasattr(reg.drr, "ncc") and fused_similarity)
        self.fused_similarity = bool(fused)

        # (two launches an iteration does not need: the sum of ONE value, and the ones autograd
        # fills in as the gradient of a scalar loss -- handed over ready-made)
        one = torch.ones((), dtype=reg._rotation.dtype, device=reg._rotation.device)

        def iteration():
            optimizer.zero_grad(set_to_none=True)
            if fused:
                values = reg.drr.ncc(target, reg._rotation, reg._translation, convention=reg.convention,
                                     eps=criterion.eps)
            else:
                values = criterion(target, reg(**render_kwargs))
            loss = values.reshape(()) if values.numel() == 1 else values.sum()
            loss.backward(gradient=one)
            optimizer.step()
            return loss.detach()

        try:
            # warm up on a side stream (first-call costs and lazy initialisations must not be
            # captured)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    iteration()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            # (the capture stream is ours, so that the fused step's zeroed workspace -- one per
            # stream -- exists before the capture: inside it, its zero-fill would become a node)
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            if fused:
                from . import ops
                with torch.cuda.stream(cap):
                    ops.siddon_ncc_workspace(reg._rotation.shape[0], reg._rotation.device)
                cap.synchronize()
            with torch.cuda.graph(self.graph, stream=cap):
                self.loss = iteration()
        finally:
            # (whatever the warm-up or the capture raised: the caller's module gets its own
            # promise back)
            if had_static is not None:
                renderer.static_volume = had_static
        # undo warm-up and capture: same parameter values, same optimizer state, IN PLACE (the
        # graph holds the addresses of both).  State the warm-up created is zeroed, which is what
        # a first step starts from (Adam's moments and step count, SGD's momentum buffer).
        with torch.no_grad():
            for p, v in zip(params, saved_params):
                p.copy_(v)
                p.grad = None
                for name, val in optimizer.state.get(p, {}).items():
                    if torch.is_tensor(val):
                        old = saved_state[p].get(name)
                        val.copy_(old) if old is not None else val.zero_()

        self._captured_hyper = self._host_hyper()

    def _host_hyper(self):
        """The param groups' hyper-parameters that live on the host (baked into the graph)."""
        return [tuple(sorted((k, v) for k, v in g.items()
                             if k != "params" and isinstance(v, (bool, int, float, tuple, type(None)))))
                for g in self.optimizer.param_groups]

    def __call__(self) -> torch.Tensor:
        if self._host_hyper() != self._captured_hyper:
            raise RuntimeError(
                "GraphedIteration: the optimizer's hyper-parameters changed after capture "
                f"({self._captured_hyper} -> {self._host_hyper()}); they are baked into the graph -- "
                "build a new GraphedIteration (or use tensor-valued learning rates)")
        self.graph.replay()
        self.iterations_done += 1
        return self.loss
