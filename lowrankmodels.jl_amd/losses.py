"""Scalar losses of LowRankModels.jl (reference: src/losses.jl), host-side descriptors.

Each class mirrors the reference constructor (same names, argument order and defaults) and
lowers to the ``glrm_loss`` descriptor of include/glrm_hip.h.  ``evaluate``/``grad`` are the
reference's scalar methods, kept on the host for API parity; the fit itself evaluates losses
inside the HIP kernels.
"""
from __future__ import annotations

import copy as _copy
import math

import numpy as np

from ._capi import LOSS_DTYPE

QUAD, L1, HUBER, QUANTILE, PERIODIC, POISSON, ORDINAL_HINGE, LOGISTIC, WEIGHTED_HINGE = range(9)


def myBool(a):
    """src/losses.jl:104-106: Int labels 1 -> true, 0 / -1 -> false, anything else throws."""
    if isinstance(a, (bool, np.bool_)):
        return bool(a)
    if a == 1:
        return True
    if a == -1 or a == 0:
        return False
    raise ValueError(f"InexactError: label {a!r} is not a Bool (expected 1, 0 or -1)")


class Loss:
    kind = -1
    classification = False  # ClassificationLoss (src/losses.jl:56)

    def __init__(self, scale=1.0):
        self.scale = float(scale)

    # mul!(l, newscale) / scale(l) / newscale * l  (src/losses.jl:61-64: `*` SETS the scale)
    def mul_(self, newscale):
        self.scale = float(newscale)
        return self

    def __rmul__(self, newscale):
        return _copy.copy(self).mul_(newscale)

    __mul__ = __rmul__

    def _params(self):
        return (0.0, 0.0)

    def descriptor(self):
        p0, p1 = self._params()
        return (self.kind, 0, self.scale, float(p0), float(p1))

    def __repr__(self):
        return f"{type(self).__name__}(scale={self.scale})"


class QuadLoss(Loss):  # src/losses.jl:138-148
    kind = QUAD

    def evaluate(self, u, a):
        return self.scale * (u - a) ** 2

    def grad(self, u, a):
        return 2 * (u - a) * self.scale


class L1Loss(Loss):  # :152-162
    kind = L1

    def evaluate(self, u, a):
        return self.scale * abs(u - a)

    def grad(self, u, a):
        return float(np.sign(u - a)) * self.scale


class HuberLoss(Loss):  # :166-179
    kind = HUBER

    def __init__(self, scale=1.0, crossover=1.0):
        super().__init__(scale)
        self.crossover = float(crossover)

    def _params(self):
        return (self.crossover, 0.0)

    def evaluate(self, u, a):
        d = abs(u - a)
        return (d - self.crossover + self.crossover ** 2) * self.scale if d > self.crossover else (u - a) ** 2 * self.scale

    def grad(self, u, a):
        return float(np.sign(u - a)) * self.scale if abs(u - a) > self.crossover else (u - a) * self.scale


class QuantileLoss(Loss):  # :186-203
    kind = QUANTILE

    def __init__(self, scale=1.0, quantile=0.5):
        super().__init__(scale)
        self.quantile = float(quantile)

    def _params(self):
        return (self.quantile, 0.0)

    def evaluate(self, u, a):
        diff = a - u
        return self.scale * self.quantile * diff if diff > 0 else -self.scale * (1 - self.quantile) * diff

    def grad(self, u, a):
        return -self.scale * self.quantile if a - u > 0 else self.scale * (1 - self.quantile)


class PeriodicLoss(Loss):  # :209-224  PeriodicLoss(T, scale=1.0)
    kind = PERIODIC

    def __init__(self, T, scale=1.0):
        super().__init__(scale)
        self.T = float(T)

    def _params(self):
        return (self.T, 0.0)

    def evaluate(self, u, a):
        return self.scale * (1 - math.cos((a - u) * (2 * math.pi) / self.T))

    def grad(self, u, a):
        return -self.scale * ((2 * math.pi) / self.T) * math.sin((a - u) * (2 * math.pi) / self.T)


class PoissonLoss(Loss):  # :231-243 (constructor fixes scale = 1.0)
    kind = POISSON

    def __init__(self, max_count=2 ** 31):
        super().__init__(1.0)
        self.max_count = max_count

    def evaluate(self, u, a):
        return self.scale * (math.exp(u) - a * u + (0 if a == 0 else a * (math.log(a) - 1)))

    def grad(self, u, a):
        return self.scale * (math.exp(u) - a)


class OrdinalHingeLoss(Loss):  # :247-294
    kind = ORDINAL_HINGE

    def __init__(self, m1=None, m2=None, scale=1.0):
        # OrdinalHingeLoss() = (1,10); OrdinalHingeLoss(m2) = (1,m2); OrdinalHingeLoss(m1,m2,scale)
        if m1 is None and m2 is None:
            m1, m2 = 1, 10
        elif m2 is None:
            m1, m2 = 1, m1
        super().__init__(scale)
        self.min, self.max = int(m1), int(m2)

    def _params(self):
        return (self.min, self.max)

    def evaluate(self, u, a):
        fl, ce = math.floor, math.ceil
        if u > self.max - 1:
            n = min(fl(u), self.max - 1) - a
            loss = n * (n + 1) / 2 + (n + 1) * (u - self.max + 1)
        elif u > a:
            n = min(fl(u), self.max) - a
            loss = n * (n + 1) / 2 + (n + 1) * (u - fl(u))
        elif u > self.min + 1:
            n = a - max(ce(u), self.min + 1)
            loss = n * (n + 1) / 2 + (n + 1) * (ce(u) - u)
        else:
            n = a - max(ce(u), self.min + 1)
            loss = n * (n + 1) / 2 + (n + 1) * (self.min + 1 - u)
        return self.scale * loss

    def grad(self, u, a):
        if u > a:
            g = min(math.ceil(u), self.max) - a
        else:
            g = -(a - max(math.floor(u), self.min))
        return self.scale * g


class LogisticLoss(Loss):  # :298-311
    kind = LOGISTIC
    classification = True

    def evaluate(self, u, a):
        a = myBool(a)
        return self.scale * math.log(1 + math.exp(-(2 * a - 1) * u))

    def grad(self, u, a):
        aa = 2 * myBool(a) - 1
        return -aa * self.scale / (1 + math.exp(aa * u))


class WeightedHingeLoss(Loss):  # :317-352
    kind = WEIGHTED_HINGE
    classification = True

    def __init__(self, scale=1.0, case_weight_ratio=1.0):
        super().__init__(scale)
        self.case_weight_ratio = float(case_weight_ratio)

    def _params(self):
        return (self.case_weight_ratio, 0.0)

    def evaluate(self, u, a):
        a = myBool(a)
        loss = self.scale * max(1 - (2 * a - 1) * u, 0)
        if self.case_weight_ratio != 1.0 and a:
            loss *= self.case_weight_ratio
        return loss

    def grad(self, u, a):
        a = myBool(a)
        an = 2 * a - 1
        g = 0 if an * u >= 1 else -an * self.scale
        if self.case_weight_ratio != 1.0 and a:
            g *= self.case_weight_ratio
        return g


def HingeLoss(scale=1.0, **kwargs):  # :323
    return WeightedHingeLoss(scale, **kwargs)


def evaluate(obj, *args):
    """evaluate(l::Loss, u, a) / evaluate(r::Regularizer, x) -- generic-function spelling."""
    return obj.evaluate(*args)


def grad(l, u, a):
    return l.grad(u, a)


def embedding_dim(losses):  # src/losses.jl:72-73 (scalar losses: 1 each)
    return 1 if isinstance(losses, Loss) else len(losses)


def pack_losses(losses):
    """list of Loss -> structured array for the ABI; a homogeneous list collapses to length 1."""
    descs = [l.descriptor() for l in losses]
    if len(set(descs)) == 1:
        descs = descs[:1]
    return np.array(descs, dtype=LOSS_DTYPE)
