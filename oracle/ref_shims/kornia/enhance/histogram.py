"""Shim: reference metrics.py:107 imports these at module level."""


def marginal_pdf(*a, **k):
    raise NotImplementedError("kornia is not available in this container")


def joint_pdf(*a, **k):
    raise NotImplementedError("kornia is not available in this container")
