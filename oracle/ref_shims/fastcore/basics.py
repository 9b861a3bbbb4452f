"""Shim for `fastcore.basics.patch` (used at reference drr.py:9, detector.py:7)."""


def patch(f):
    # `from __future__ import annotations` makes the annotation a string.
    owner = f.__annotations__["self"]
    cls = eval(owner, f.__globals__) if isinstance(owner, str) else owner
    setattr(cls, f.__name__, f)
    return f
