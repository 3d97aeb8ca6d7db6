"""Regularizers named by the north star (reference: src/regularizers.jl), host descriptors.

``evaluate``/``prox`` restate the reference's array methods for API parity; inside the fit the
prox step runs fused in the HIP sweep kernels.
"""
from __future__ import annotations

import copy as _copy

import numpy as np

from ._capi import REG_DTYPE, bump_epoch as _bump_epoch

ZERO, QUAD, ONE, NONNEG, UNIT_ONE_SPARSE = range(5)


class Regularizer:
    kind = -1

    def __setattr__(self, name, value):  # any change of a descriptor field invalidates cached packed descriptors
        object.__setattr__(self, name, value)
        _bump_epoch()

    def __init__(self, scale=1.0):
        self.scale = float(scale)

    def mul_(self, newscale):  # mul!(r, newscale), src/regularizers.jl:38
        self.scale = float(newscale)
        return self

    def __rmul__(self, newscale):  # *(newscale, r): scale(r)*newscale on a fresh copy, :40
        r = _copy.copy(self)
        r.mul_(self.scale * newscale)
        return r

    wrap = 0

    def descriptor(self):
        return (self.kind, self.wrap, self.scale)

    def __repr__(self):
        return f"{type(self).__name__}({self.scale})"


class QuadReg(Regularizer):  # :52-58
    kind = QUAD

    def __init__(self, scale=1):
        super().__init__(scale)

    def evaluate(self, a):
        return self.scale * float(np.sum(np.abs(np.asarray(a, dtype=float)) ** 2))

    def prox(self, u, alpha):
        return 1 / (1 + 2 * alpha * self.scale) * np.asarray(u, dtype=float)


class OneReg(Regularizer):  # :79-88
    kind = ONE

    def __init__(self, scale=1):
        super().__init__(scale)

    def evaluate(self, a):
        return self.scale * float(np.sum(np.abs(a)))

    def prox(self, u, alpha):
        u = np.asarray(u, dtype=float)
        t = self.scale * alpha
        return np.maximum(u - t, 0) + np.minimum(u + t, 0)


class _Unscaled(Regularizer):
    def __init__(self):
        super().__init__(1.0)

    def mul_(self, newscale):  # mul!(r::ZeroReg/NonNeg/UnitOneSparse, _) is a no-op (:97,:114,:318)
        return self

    def descriptor(self):
        return (self.kind, 0, 1.0)

    def __repr__(self):
        return f"{type(self).__name__}()"


class ZeroReg(_Unscaled):  # :91-97
    kind = ZERO

    def evaluate(self, a):
        return 0

    def prox(self, u, alpha):
        return np.asarray(u, dtype=float)


class NonNegConstraint(_Unscaled):  # :101-114
    kind = NONNEG

    def evaluate(self, a):
        return float("inf") if np.any(np.asarray(a) < 0) else 0

    def prox(self, u, alpha=1):
        return np.maximum(np.asarray(u, dtype=float), 0)


class UnitOneSparseConstraint(_Unscaled):  # :295-318
    kind = UNIT_ONE_SPARSE

    def evaluate(self, a):
        oneflag = False
        for ai in np.asarray(a).ravel():
            if ai == 0:
                continue
            if ai == 1:
                if oneflag:
                    return float("inf")
                oneflag = True
            else:
                return float("inf")
        return 0

    def prox(self, u, alpha=0):
        u = np.asarray(u, dtype=float)
        v = np.zeros_like(u)
        v[int(np.argmax(u))] = 1
        return v


# ------------------------------------------------------------------------- wrappers / block regularizers
WRAP_LASTENTRY1, WRAP_LASTENTRY_UNPENALIZED, WRAP_ORDINAL, WRAP_MNL_ORDINAL = 1, 2, 4, 8


class _Wrapper(Regularizer):
    """A regularizer around a base regularizer r (one of the five above).  Arrays are k-vectors or k x d blocks
    (first axis = latent component), like the views the reference passes."""
    wrap = 0

    def __init__(self, r=None):
        r = ZeroReg() if r is None else r
        if isinstance(r, _Wrapper) or r.kind < 0:
            raise NotImplementedError("nested wrappers are outside the accelerated path")
        self.r = r

    kind = property(lambda self: self.r.kind)
    scale = property(lambda self: self.r.scale)

    def mul_(self, newscale):
        self.r.mul_(newscale)
        return self

    def descriptor(self):
        return (self.r.kind, self.wrap, self.r.descriptor()[2])

    def __repr__(self):
        return f"{type(self).__name__}({self.r!r})"


class lastentry1(_Wrapper):  # src/regularizers.jl:163-175
    wrap = WRAP_LASTENTRY1

    def evaluate(self, a):
        a = np.asarray(a, dtype=float)
        return self.r.evaluate(a[:-1]) if np.all(a[-1] == 1) else float("inf")

    def prox(self, u, alpha=1):
        u = np.array(u, dtype=float)
        u[:-1] = self.r.prox(u[:-1], alpha)
        u[-1] = 1
        return u


class lastentry_unpenalized(_Wrapper):  # src/regularizers.jl:177-189
    wrap = WRAP_LASTENTRY_UNPENALIZED

    def __new__(cls, r=None):
        if isinstance(r, (OrdinalReg, MNLOrdinalReg)):  # "make sure we don't add two offsets", :386,:411
            return r
        return super().__new__(cls)

    def evaluate(self, a):
        return self.r.evaluate(np.asarray(a, dtype=float)[:-1])

    def prox(self, u, alpha=1):
        u = np.array(u, dtype=float)
        u[:-1] = self.r.prox(u[:-1], alpha)
        return u


class OrdinalReg(_Wrapper):  # src/regularizers.jl:356-386
    wrap = WRAP_ORDINAL

    def evaluate(self, a):
        a = np.asarray(a, dtype=float).reshape(len(a), -1)
        return self.r.evaluate(a[:-1, 0])

    def prox(self, u, alpha):
        u = np.array(u, dtype=float)
        u2 = u.reshape(len(u), -1)
        um = np.asarray(self.r.prox(np.mean(u2[:-1, :], axis=1), alpha), dtype=float)
        u2[:-1, :] = um[:, None]
        return u


class MNLOrdinalReg(OrdinalReg):  # src/regularizers.jl:388-411
    wrap = WRAP_MNL_ORDINAL

    def prox(self, u, alpha, TOL=1e-3):
        u = OrdinalReg.prox(self, u, alpha)
        u2 = u.reshape(len(u), -1)
        u2[-1, 0] = min(-TOL, u2[-1, 0])
        for j in range(1, u2.shape[1]):
            u2[-1, j] = min(u2[-1, j], u2[-1, j - 1] - TOL)
        return u


def prox(r, u, alpha):
    return r.prox(u, alpha)


def pack_regs(regs):
    descs = [r.descriptor() for r in regs]
    if len(set(descs)) == 1:
        descs = descs[:1]
    return np.array(descs, dtype=REG_DTYPE)
