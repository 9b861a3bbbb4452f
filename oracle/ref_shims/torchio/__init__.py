"""Shim for `torchio.Subject` (used at reference drr.py:18): a plain attribute bag."""


class _Image:
    def __init__(self, data, affine=None):
        self.data = data
        self.affine = affine


class ScalarImage(_Image):
    pass


class LabelMap(_Image):
    pass


class Subject:
    def __init__(self, **kw):
        self.__dict__.update(kw)
