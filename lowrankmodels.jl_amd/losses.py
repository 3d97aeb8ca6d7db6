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

from ._capi import LOSS_DTYPE, bump_epoch as _bump_epoch

QUAD, L1, HUBER, QUANTILE, PERIODIC, POISSON, ORDINAL_HINGE, LOGISTIC, WEIGHTED_HINGE = range(9)
MULTINOMIAL, OVA, BVS, ORDISTIC, MULTINOMIAL_ORDINAL = range(9, 14)
MAX_EMBEDDING_DIM = 32


def _exp(x):
    """exp with IEEE semantics (Julia's exp overflows to Inf instead of raising)."""
    try:
        return math.exp(x)
    except OverflowError:
        return math.inf


def _log(x):
    """log with IEEE semantics: log(0) = -Inf, log(negative) = NaN, log(Inf) = Inf."""
    if x > 0:
        return math.log(x) if x != math.inf else math.inf
    return -math.inf if x == 0 else math.nan


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

    def __setattr__(self, name, value):  # any change of a descriptor field invalidates cached packed descriptors
        object.__setattr__(self, name, value)
        _bump_epoch()

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

    embedding_dim = 1  # src/losses.jl:72

    def descriptor(self):
        p0, p1 = self._params()
        return (self.kind, 0 if self.embedding_dim == 1 else int(self.embedding_dim), self.scale, float(p0), float(p1))

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
        return self.scale * (_exp(u) - a * u + (0 if a == 0 else a * (_log(a) - 1)))

    def grad(self, u, a):
        return self.scale * (_exp(u) - a)


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
        return self.scale * _log(1 + _exp(-(2 * a - 1) * u))

    def grad(self, u, a):
        aa = 2 * myBool(a) - 1
        return -aa * self.scale / (1 + _exp(aa * u))


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


# ---------------------------------------------------------------------------------- multi-dimensional losses
# The column owns `embedding_dim` consecutive columns of Y; u is the vector x_e' Y[:, yidxs[f]] and a the level 1..max
# (datalevels, src/losses.jl:367,422,459,497,570).

class _MultiDimLoss(Loss):
    def __init__(self, max_, scale=1.0):
        super().__init__(scale)
        self.max = int(max_)
        if not 2 <= self.embedding_dim <= MAX_EMBEDDING_DIM:
            raise ValueError(f"embedding dimension {self.embedding_dim} outside 2..{MAX_EMBEDDING_DIM}")

    def __repr__(self):
        return f"{type(self).__name__}({self.max}, scale={self.scale})"


class MultinomialLoss(_MultiDimLoss):  # src/losses.jl:360-409
    kind = MULTINOMIAL
    embedding_dim = property(lambda self: self.max)

    def evaluate(self, u, a):
        u = np.asarray(u, dtype=float)
        a = int(a) - 1
        M = np.max(u) - u[a]
        sumexp = 0.0
        for j in range(len(u)):
            sumexp += _exp(u[j] - u[a] - M)
        return self.scale * (_log(sumexp) + M)

    def grad(self, u, a):
        u = np.asarray(u, dtype=float)
        g = np.zeros(len(u))
        g[int(a) - 1] = -1
        for j in range(len(u)):
            M = np.max(u) - u[j]
            sumexp = 0.0
            for jp in range(len(u)):
                sumexp += _exp(u[jp] - u[j] - M)
            g[j] += _exp(-M) / sumexp
        return self.scale * g


class _BinWrapped(_MultiDimLoss):
    def __init__(self, max_, scale=1.0, bin_loss=None):
        self.max = int(max_)
        Loss.__init__(self, scale)
        self.bin_loss = LogisticLoss(scale) if bin_loss is None else bin_loss  # bin_loss=LogisticLoss(scale), :420,:457
        if not (isinstance(self.bin_loss, LogisticLoss) or (isinstance(self.bin_loss, WeightedHingeLoss) and self.bin_loss.case_weight_ratio == 1.0)):
            raise NotImplementedError("bin_loss must be LogisticLoss or HingeLoss")
        if not 2 <= self.embedding_dim <= MAX_EMBEDDING_DIM:
            raise ValueError(f"embedding dimension {self.embedding_dim} outside 2..{MAX_EMBEDDING_DIM}")

    def _params(self):
        return (self.bin_loss.scale, self.bin_loss.kind)


class OvALoss(_BinWrapped):  # src/losses.jl:413-446
    kind = OVA
    embedding_dim = property(lambda self: self.max)

    def evaluate(self, u, a):
        return self.scale * sum(self.bin_loss.evaluate(u[j], int(a) == j + 1) for j in range(len(u)))

    def grad(self, u, a):
        return self.scale * np.array([self.bin_loss.grad(u[j], int(a) == j + 1) for j in range(len(u))], dtype=float)


class BvSLoss(_BinWrapped):  # src/losses.jl:450-483
    kind = BVS
    embedding_dim = property(lambda self: self.max - 1)

    def evaluate(self, u, a):
        return self.scale * sum(self.bin_loss.evaluate(u[j], int(a) > j + 1) for j in range(len(u)))

    def grad(self, u, a):
        return self.scale * np.array([self.bin_loss.grad(u[j], int(a) > j + 1) for j in range(len(u))], dtype=float)


class OrdisticLoss(_MultiDimLoss):  # src/losses.jl:490-530
    kind = ORDISTIC
    embedding_dim = property(lambda self: self.max)

    def evaluate(self, u, a):
        u = np.asarray(u, dtype=float)
        diffusquared = u[int(a) - 1] ** 2 - u ** 2
        M = np.max(diffusquared)
        invlik = float(np.sum(np.exp(diffusquared - M)))
        return self.scale * (M + _log(invlik))

    def grad(self, u, a):
        u = np.asarray(u, dtype=float)
        g = np.zeros(len(u))
        g[int(a) - 1] = 2 * u[int(a) - 1]
        for j in range(len(u)):
            diffusquared = u[j] ** 2 - u ** 2
            M = np.max(diffusquared)
            invlik = float(np.sum(np.exp(diffusquared - M)))
            g[j] -= 2 * u[j] * _exp(-M) / invlik
        return self.scale * g


def enforce_MNLOrdRules(u, TOL=1e-3):  # src/losses.jl:572-578 (on a copy: XY[e, range] is a copy in the reference)
    u = np.array(u, dtype=float)
    u[0] = min(-TOL, u[0])
    for j in range(1, len(u)):
        u[j] = min(u[j], u[j - 1] - TOL)
    return u


class MultinomialOrdinalLoss(_MultiDimLoss):  # src/losses.jl:562-620
    kind = MULTINOMIAL_ORDINAL
    embedding_dim = property(lambda self: self.max - 1)

    def evaluate(self, u, a):
        u, a = enforce_MNLOrdRules(u), int(a)
        if a == 1:
            return -self.scale * _log(_exp(0) - _exp(u[0]))
        if a == self.max:
            return -self.scale * u[a - 2]
        return -self.scale * _log(_exp(u[a - 2]) - _exp(u[a - 1]))

    def grad(self, u, a):
        u, a = enforce_MNLOrdRules(u), int(a)
        g = np.zeros(len(u))
        if a == 1:
            g[0] = -_exp(u[0]) / (_exp(0) - _exp(u[0]))
        elif a == self.max:
            g[a - 2] = 1
        else:
            g[a - 1] = -_exp(u[a - 1]) / (_exp(u[a - 2]) - _exp(u[a - 1]))
            g[a - 2] = _exp(u[a - 2]) / (_exp(u[a - 2]) - _exp(u[a - 1]))
        return -self.scale * g


def evaluate(obj, *args):
    """evaluate(l::Loss, u, a) / evaluate(r::Regularizer, x) -- generic-function spelling."""
    return obj.evaluate(*args)


def grad(l, u, a):
    return l.grad(u, a)


def embedding_dim(losses):  # src/losses.jl:72-73
    return losses.embedding_dim if isinstance(losses, Loss) else sum(l.embedding_dim for l in losses)


def get_yidxs(losses):
    """Column spans of Y per column of A (src/losses.jl:76-93), 0-based half-open (start, stop)."""
    out, start = [], 0
    for l in losses:
        out.append((start, start + l.embedding_dim))
        start += l.embedding_dim
    return out


def pack_losses(losses):
    """list of Loss -> structured array for the ABI; a homogeneous list collapses to length 1."""
    descs = [l.descriptor() for l in losses]
    if len(set(descs)) == 1:
        descs = descs[:1]
    return np.array(descs, dtype=LOSS_DTYPE)
