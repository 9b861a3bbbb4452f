"""SE(3) camera poses: ``RigidTransform`` and ``convert``.

API-compatible restatement of the feeder layer the renderers sit behind
(reference ``diffdrr/pose.py:14-190`` for ``RigidTransform`` / ``convert``;
``pose.py:193-253`` 9d / 10d / quaternion-adjugate; the rotation conversions the
reference vendors from pytorch3d at ``pose.py:256-1333``).  It stays plain
PyTorch on purpose: B x 4 x 4 algebra is negligible next to rendering and
autograd carries the SE(3) chain rule down to the ray endpoints the HIP kernels
differentiate.

Conventions: a pose built from ``(rotation, translation)``
is the 4x4 matrix ``[R | R t]``, i.e. the camera centre is ``R @ t``
(reference pose.py:155-157); ``a.compose(b)`` is ``b.matrix @ a.matrix``
(pose.py:69-71); points are column vectors.
"""
from __future__ import annotations

import math

import torch

PARAMETERIZATIONS = [
    "axis_angle",
    "euler_angles",
    "matrix",
    "quaternion",
    "quaternion_adjugate",
    "rotation_6d",
    "rotation_9d",
    "rotation_10d",
    "se3_log_map",
]

_AXIS = {"X": 0, "Y": 1, "Z": 2}


def _is_orthonormal(R: torch.Tensor, eps: float) -> bool:
    eye = torch.eye(3, dtype=R.dtype, device=R.device)
    return bool(torch.all(torch.linalg.matrix_norm(R @ R.mT - eye) < eps))


class RigidTransform(torch.nn.Module):
    """A batch of rigid (or affine) 4x4 transforms acting on point clouds.

    Mirrors reference pose.py:14-105: ``forward`` applies the top 3x4 block to
    ``(B, N, 3)`` points, ``inverse`` uses the closed form when the rotation
    block is orthonormal, ``compose`` left-multiplies, ``convert`` returns a
    ``(rotation, translation)`` pair in any supported parameterization.
    """

    def __new__(cls, matrix, eps=1e-6):
        if isinstance(matrix, cls):
            return matrix
        return super().__new__(cls)

    def __init__(self, matrix, eps=1e-6):
        if isinstance(matrix, type(self)):
            return
        super().__init__()
        if matrix.dim() == 2:
            matrix = matrix.unsqueeze(0)
        self.register_buffer("matrix", matrix)
        self.eps = eps

    def __len__(self):
        return len(self.matrix)

    def __getitem__(self, idx):
        return type(self)(self.matrix[idx])

    def __matmul__(self, other):
        return other.compose(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Apply to points ``(B, N, 3)`` (or ``(1, N, 3)``, broadcast over the batch)."""
        A = self.matrix[:, :3, :3]
        b = self.matrix[:, :3, 3]
        return torch.einsum("bij,bnj->bni", A, x) + b.unsqueeze(1)

    @property
    def rotation(self):
        return self.matrix[..., :3, :3]

    @property
    def translation(self):
        return self.matrix[..., :3, 3]

    def inverse(self) -> "RigidTransform":
        R = self.rotation
        if _is_orthonormal(R, self.eps):
            Rinv = R.mT
            tinv = -torch.einsum("bij,bj->bi", Rinv, self.translation)
            return type(self)(make_matrix(Rinv, tinv))
        return type(self)(torch.linalg.inv(self.matrix))

    def compose(self, other: "RigidTransform") -> "RigidTransform":
        return type(self)(torch.einsum("bij,bjk->bik", other.matrix, self.matrix))

    def convert(self, parameterization, convention=None, degrees=False):
        R = self.rotation
        translation = -self.inverse().translation
        if parameterization == "axis_angle":
            rotation = matrix_to_axis_angle(R)
        elif parameterization == "euler_angles":
            rotation = matrix_to_euler_angles(R, convention)
            if degrees:
                rotation = rotation / math.pi * 180
        elif parameterization == "matrix":
            rotation = R
        elif parameterization == "quaternion":
            rotation = standardize_quaternion(matrix_to_quaternion(R))
        elif parameterization == "quaternion_adjugate":
            rotation = quaternion_to_quaternion_adjugate(matrix_to_quaternion(R))
        elif parameterization == "rotation_6d":
            rotation = matrix_to_rotation_6d(R)
        elif parameterization == "rotation_9d":
            rotation = matrix_to_rotation_9d(R)
        elif parameterization == "rotation_10d":
            rotation = quaternion_to_rotation_10d(matrix_to_quaternion(R))
        elif parameterization == "se3_log_map":
            params = self.get_se3_log()
            rotation, translation = params[..., 3:], params[..., :3]
        else:
            raise ValueError(f"Must be in {PARAMETERIZATIONS}, not {parameterization}")
        return rotation, translation

    def get_se3_log(self):
        return se3_log_map(self.matrix)


class _PoseEulerFn(torch.autograd.Function):
    """Euler angles (radians) + translation -> world pose of the C-arm, (B,3,4) =
    ([R | R xyz] @ reorient)[:, :3] as ONE kernel each way (ddrr_pose_euler_forward /
    _backward) instead of the ~35 + ~70 small launches of `convert` + `compose` below:
    with the renderer at ~0.3 ms per DRR those launches were most of a registration step."""

    @staticmethod
    def forward(ctx, rot, xyz, reorient34, axes):
        from . import ops

        ctx.axes = axes
        ctx.save_for_backward(rot, xyz, reorient34)
        return ops.pose_euler_forward(rot, xyz, axes, reorient34)

    @staticmethod
    def backward(ctx, gMw):
        from . import ops

        rot, xyz, reorient34 = ctx.saved_tensors
        g_rot, g_xyz = ops.pose_euler_backward(rot, xyz, ctx.axes, reorient34, gMw)
        return g_rot, g_xyz, None, None


def euler_world_pose(rot, xyz, convention, reorient, degrees=False):
    """``(reorient.compose(convert(rot, xyz, "euler_angles", convention))).matrix[:, :3]``
    through the fused kernels (float32 tensors on the GPU)."""
    _check_convention(convention)
    if degrees:
        rot = rot / 180 * math.pi
    axes = tuple(_AXIS[c] for c in convention)
    return _PoseEulerFn.apply(rot, xyz, reorient[:3, :].contiguous(), axes)


def make_matrix(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    assert len(R) == len(t)
    bottom = torch.zeros(len(R), 1, 4, dtype=R.dtype, device=R.device)
    bottom[..., 0, 3] = 1.0
    return torch.cat([torch.cat([R, t.unsqueeze(-1)], dim=-1), bottom], dim=-2)


def random_rigid_transform(batch_size=1, generator=None) -> RigidTransform:
    """Uniformly random rotations + N(0, 100^2) translations (testing helper)."""
    q = torch.randn(batch_size, 4, generator=generator)
    t = 100 * torch.randn(batch_size, 3, generator=generator)
    return RigidTransform(make_matrix(quaternion_to_matrix(q), t))


def convert(*args, parameterization, convention=None, degrees=False) -> RigidTransform:
    """Build a ``RigidTransform`` from a rotation parameterization + translation.

    Same contract as reference pose.py:140-190."""
    if parameterization == "euler_angles" and convention is None:
        raise ValueError(
            "convention for Euler angles must be specified as a 3 letter combination of [X, Y, Z]"
        )
    if parameterization == "matrix":
        return RigidTransform(args[0])
    if parameterization == "se3_log_map":
        rotation, translation = args
        return RigidTransform(se3_exp_map(torch.cat([translation, rotation], dim=-1)))
    if parameterization not in PARAMETERIZATIONS:
        raise ValueError(f"Must be in {PARAMETERIZATIONS}, not {parameterization}")

    rotation, translation = args
    if parameterization == "axis_angle":
        R = axis_angle_to_matrix(rotation)
    elif parameterization == "euler_angles":
        if degrees:
            rotation = rotation / 180 * math.pi
        R = euler_angles_to_matrix(rotation, convention)
    elif parameterization == "quaternion":
        R = quaternion_to_matrix(rotation)
    elif parameterization == "quaternion_adjugate":
        R = quaternion_to_matrix(quaternion_adjugate_to_quaternion(rotation))
    elif parameterization == "rotation_6d":
        R = rotation_6d_to_matrix(rotation)
    elif parameterization == "rotation_9d":
        R = rotation_9d_to_matrix(rotation)
    else:  # rotation_10d
        R = quaternion_to_matrix(rotation_10d_to_quaternion(rotation))
    centre = torch.einsum("bij,bj->bi", R, translation)
    return RigidTransform(make_matrix(R, centre))


# --------------------------------------------------------------- rotations


def _elementary_rotation(axis: str, angle: torch.Tensor) -> torch.Tensor:
    c, s = torch.cos(angle), torch.sin(angle)
    one, zero = torch.ones_like(angle), torch.zeros_like(angle)
    rows = {
        "X": (one, zero, zero, zero, c, -s, zero, s, c),
        "Y": (c, zero, s, zero, one, zero, -s, zero, c),
        "Z": (c, -s, zero, s, c, zero, zero, zero, one),
    }[axis]
    return torch.stack(rows, dim=-1).reshape(angle.shape + (3, 3))


def _check_convention(convention: str):
    if (
        not isinstance(convention, str)
        or len(convention) != 3
        or any(c not in _AXIS for c in convention)
        or convention[1] in (convention[0], convention[2])
    ):
        raise ValueError(f"Invalid convention {convention}.")


def euler_angles_to_matrix(euler_angles: torch.Tensor, convention: str) -> torch.Tensor:
    """``R = R_c0(a0) @ R_c1(a1) @ R_c2(a2)`` (intrinsic rotations, radians)."""
    if euler_angles.dim() == 0 or euler_angles.shape[-1] != 3:
        raise ValueError("Invalid input euler angles.")
    _check_convention(convention)
    a = euler_angles.unbind(-1)
    R0, R1, R2 = (_elementary_rotation(c, x) for c, x in zip(convention, a))
    return R0 @ R1 @ R2


def matrix_to_euler_angles(matrix: torch.Tensor, convention: str) -> torch.Tensor:
    """Inverse of :func:`euler_angles_to_matrix` (middle angle in [-pi/2, pi/2]
    for Tait-Bryan conventions, [0, pi] for proper Euler ones)."""
    _check_convention(convention)
    i, j, k = (_AXIS[c] for c in convention)
    R = matrix
    if i != k:  # Tait-Bryan
        sigma = 1.0 if (j - i) % 3 == 1 else -1.0
        first = torch.atan2(-sigma * R[..., j, k], R[..., k, k])
        middle = torch.asin(sigma * R[..., i, k])
        last = torch.atan2(-sigma * R[..., i, j], R[..., i, i])
    else:  # proper Euler: R_i(a) R_j(b) R_i(c)
        m = 3 - i - j
        sigma = 1.0 if (j - i) % 3 == 1 else -1.0
        first = torch.atan2(R[..., j, i], -sigma * R[..., m, i])
        middle = torch.acos(R[..., i, i])
        last = torch.atan2(R[..., i, j], sigma * R[..., i, m])
    return torch.stack([first, middle, last], dim=-1)


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    """Real-part-first quaternions (not necessarily unit) -> rotation matrices."""
    w, x, y, z = quaternions.unbind(-1)
    s = 2.0 / (quaternions * quaternions).sum(-1)
    rows = (
        1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
        s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
        s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y),
    )
    return torch.stack(rows, dim=-1).reshape(quaternions.shape[:-1] + (3, 3))


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """Rotation matrices -> unit quaternions (real part first), picking per
    matrix the best-conditioned of the four classical candidates."""
    if matrix.shape[-2:] != (3, 3):
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    m = matrix
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    # 4 q_a^2 for a = w, x, y, z
    four_sq = torch.stack(
        [1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1
    )
    two_abs = torch.sqrt(four_sq.clamp_min(0.0))  # 2 |q_a|
    # candidate c holds 2 q_c * (w, x, y, z)
    cand = torch.stack(
        [
            torch.stack([four_sq[..., 0], m[..., 2, 1] - m[..., 1, 2],
                         m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1]], -1),
            torch.stack([m[..., 2, 1] - m[..., 1, 2], four_sq[..., 1],
                         m[..., 1, 0] + m[..., 0, 1], m[..., 0, 2] + m[..., 2, 0]], -1),
            torch.stack([m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] + m[..., 0, 1],
                         four_sq[..., 2], m[..., 1, 2] + m[..., 2, 1]], -1),
            torch.stack([m[..., 1, 0] - m[..., 0, 1], m[..., 2, 0] + m[..., 0, 2],
                         m[..., 2, 1] + m[..., 1, 2], four_sq[..., 3]], -1),
        ],
        dim=-2,
    )
    cand = cand / (2.0 * two_abs.unsqueeze(-1).clamp_min(0.1))
    best = two_abs.argmax(dim=-1)
    idx = best[..., None, None].expand(best.shape + (1, 4))
    return standardize_quaternion(cand.gather(-2, idx).squeeze(-2))


def standardize_quaternion(quaternions: torch.Tensor) -> torch.Tensor:
    """Flip so that the real part is non-negative."""
    return torch.where(quaternions[..., 0:1] < 0, -quaternions, quaternions)


def axis_angle_to_quaternion(axis_angle: torch.Tensor) -> torch.Tensor:
    angle = torch.linalg.vector_norm(axis_angle, dim=-1, keepdim=True)
    half = 0.5 * angle
    small = angle.abs() < 1e-6
    # sin(a/2)/a with its Taylor expansion near 0
    k = torch.where(small, 0.5 - angle * angle / 48, torch.sin(half) / torch.where(small, 1.0, angle))
    return torch.cat([torch.cos(half), axis_angle * k], dim=-1)


def quaternion_to_axis_angle(quaternions: torch.Tensor) -> torch.Tensor:
    norm = torch.linalg.vector_norm(quaternions[..., 1:], dim=-1, keepdim=True)
    half = torch.atan2(norm, quaternions[..., 0:1])
    angle = 2 * half
    small = angle.abs() < 1e-6
    k = torch.where(small, 0.5 - angle * angle / 48, torch.sin(half) / torch.where(small, 1.0, angle))
    return quaternions[..., 1:] / k


def axis_angle_to_matrix(axis_angle: torch.Tensor) -> torch.Tensor:
    return quaternion_to_matrix(axis_angle_to_quaternion(axis_angle))


def matrix_to_axis_angle(matrix: torch.Tensor) -> torch.Tensor:
    return quaternion_to_axis_angle(matrix_to_quaternion(matrix))


def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    """Gram-Schmidt on the two 3-vectors (Zhou et al. 2019); rows b1, b2, b1 x b2."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack([b1, b2, torch.cross(b1, b2, dim=-1)], dim=-2)


def matrix_to_rotation_6d(matrix: torch.Tensor) -> torch.Tensor:
    return matrix[..., :2, :].clone().reshape(matrix.shape[:-2] + (6,))


def rotation_9d_to_matrix(rotation: torch.Tensor) -> torch.Tensor:
    """Symmetric orthogonalisation of a 9-vector onto SO(3) via SVD."""
    m = rotation.reshape(-1, 3, 3)
    u, _, vh = torch.linalg.svd(m)
    det = torch.linalg.det(u @ vh).reshape(-1, 1, 1)
    vh = torch.cat([vh[:, :2], det * vh[:, 2:]], dim=1)
    return u @ vh


def matrix_to_rotation_9d(matrix: torch.Tensor) -> torch.Tensor:
    return matrix.flatten(start_dim=1)


def _sym4(vec: torch.Tensor) -> torch.Tensor:
    A = torch.zeros(len(vec), 4, 4, dtype=vec.dtype, device=vec.device)
    iu = torch.triu_indices(4, 4)
    A[:, iu[0], iu[1]] = vec
    A[:, iu[1], iu[0]] = vec
    return A


def rotation_10d_to_quaternion(rotation: torch.Tensor) -> torch.Tensor:
    """Eigenvector of the smallest eigenvalue of the symmetric 4x4 built from a
    10-vector (Peretroukhin et al. 2020)."""
    return torch.linalg.eigh(_sym4(rotation)).eigenvectors[..., 0]


def quaternion_to_rotation_10d(q: torch.Tensor) -> torch.Tensor:
    iu = torch.triu_indices(4, 4)
    return (-torch.einsum("bi,bj->bij", q, q))[:, iu[0], iu[1]]


def quaternion_adjugate_to_quaternion(rotation: torch.Tensor) -> torch.Tensor:
    """Column of largest norm of the quaternion adjugate, scaled by that norm
    (Lin et al. 2022): an un-normalised quaternion."""
    A = _sym4(rotation)
    norms = torch.linalg.vector_norm(A, dim=1)
    col = norms.argmax(dim=1)
    return A[torch.arange(len(A)), col] / norms.amax(dim=1, keepdim=True)


def quaternion_to_quaternion_adjugate(q: torch.Tensor) -> torch.Tensor:
    iu = torch.triu_indices(4, 4)
    return torch.einsum("bi,bj->bij", q, q)[:, iu[0], iu[1]]


# ------------------------------------------------------------- se(3) maps


def _hat(v: torch.Tensor) -> torch.Tensor:
    x, y, z = v.unbind(-1)
    o = torch.zeros_like(x)
    return torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=-1).reshape(v.shape[:-1] + (3, 3))


def _so3_coefficients(theta: torch.Tensor, eps: float):
    """sin(t)/t, (1-cos t)/t^2, (t - sin t)/t^3 with the angle clamped away from 0."""
    t = theta.clamp_min(eps)
    return torch.sin(t) / t, (1 - torch.cos(t)) / (t * t), (t - torch.sin(t)) / (t * t * t)


def se3_exp_map(log_transform: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """``[u, w] -> [[exp(hat w), V u], [0, 1]]`` (column-vector convention)."""
    u, w = log_transform[..., :3], log_transform[..., 3:]
    theta = torch.sqrt((w * w).sum(-1).clamp_min(eps * eps))
    a, b, c = _so3_coefficients(theta, eps)
    K = _hat(w)
    K2 = K @ K
    eye = torch.eye(3, dtype=w.dtype, device=w.device)
    R = eye + a[..., None, None] * K + b[..., None, None] * K2
    V = eye + b[..., None, None] * K + c[..., None, None] * K2
    return make_matrix(R, torch.einsum("bij,bj->bi", V, u))


def se3_log_map(matrix: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """Inverse of :func:`se3_exp_map`; returns ``[u, w]``."""
    R, t = matrix[..., :3, :3], matrix[..., :3, 3]
    w = matrix_to_axis_angle(R)
    theta = torch.sqrt((w * w).sum(-1).clamp_min(eps * eps))
    _, b, c = _so3_coefficients(theta, eps)
    K = _hat(w)
    eye = torch.eye(3, dtype=R.dtype, device=R.device)
    V = eye + b[..., None, None] * K + c[..., None, None] * (K @ K)
    u = torch.linalg.solve(V, t.unsqueeze(-1)).squeeze(-1)
    return torch.cat([u, w], dim=-1)
