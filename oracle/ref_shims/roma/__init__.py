"""Shim for `roma.is_orthonormal_matrix` (used at reference pose.py:11, 59)."""
import torch


def is_orthonormal_matrix(R, epsilon=1e-7):
    eye = torch.eye(R.shape[-1], dtype=R.dtype, device=R.device)
    return bool(torch.all(torch.linalg.norm(R @ R.mT - eye, dim=(-1, -2)) < epsilon))
