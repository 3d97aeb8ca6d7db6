"""Domains and post-fit evaluation (reference: src/domains.jl, src/impute_and_err.jl, src/evaluate_fit.jl:107-168): how a column's
values are imputed from the fitted model, and the error metrics built on that.  ``impute_entry`` / ``error_metric_entry`` are
the scalar host mirrors; ``impute(glrm)`` and ``error_metric(glrm, ...)`` run on the engine from the model's resident handle."""
from __future__ import annotations

import math

import numpy as np

from . import _capi
from . import losses as _l

REAL, BOOL, ORDINAL, PERIODIC, COUNT, CATEGORICAL = range(6)


class Domain:
    kind = -1

    def descriptor(self):
        return (self.kind, 0, 0.0, 0.0)

    def __repr__(self):
        return f"{type(self).__name__}()"


class RealDomain(Domain):
    kind = REAL


class BoolDomain(Domain):
    kind = BOOL


class OrdinalDomain(Domain):
    kind = ORDINAL

    def __init__(self, min_, max_):
        self.min, self.max = int(min_), int(max_)

    def descriptor(self):
        return (self.kind, 0, float(self.min), float(self.max))

    def __repr__(self):
        return f"OrdinalDomain({self.min}, {self.max})"


class CategoricalDomain(Domain):
    kind = CATEGORICAL

    def __init__(self, max_):
        self.min, self.max = 1, int(max_)

    def descriptor(self):
        return (self.kind, 0, 1.0, float(self.max))

    def __repr__(self):
        return f"CategoricalDomain({self.max})"


class PeriodicDomain(Domain):
    kind = PERIODIC

    def __init__(self, T):
        self.T = float(T)

    def descriptor(self):
        return (self.kind, 0, self.T, 0.0)

    def __repr__(self):
        return f"PeriodicDomain({self.T})"


class CountDomain(Domain):
    kind = COUNT

    def __init__(self, max_count):
        self.max_count = int(max_count)

    def descriptor(self):
        return (self.kind, 0, 0.0, float(self.max_count))

    def __repr__(self):
        return f"CountDomain({self.max_count})"


def default_domain(loss):
    """l.domain of the reference constructors (src/losses.jl:142-567)."""
    if isinstance(loss, _l.PeriodicLoss):
        return PeriodicDomain(loss.T)
    if isinstance(loss, _l.PoissonLoss):
        return CountDomain(loss.max_count)
    if isinstance(loss, _l.OrdinalHingeLoss):
        return OrdinalDomain(loss.min, loss.max)
    if loss.classification:
        return BoolDomain()
    if isinstance(loss, (_l.MultinomialLoss, _l.OvALoss)):
        return CategoricalDomain(loss.max)
    if isinstance(loss, (_l.BvSLoss, _l.OrdisticLoss, _l.MultinomialOrdinalLoss)):
        return OrdinalDomain(1, loss.max)
    return RealDomain()


def pack_domains(domains):
    return np.array([d.descriptor() for d in domains], dtype=_capi.DOMAIN_DTYPE)


def roundcutoff(x, a, b):
    """T(min(max(round(x), a), b)) -- Julia's round is half-to-even, like Python's."""
    return min(max(float(round(x)), a), b)


_DIFF = (_l.QuadLoss, _l.L1Loss, _l.HuberLoss, _l.QuantileLoss, _l.PeriodicLoss)


def impute_entry(D, l, u):
    """impute(D, l, u) (src/impute_and_err.jl:37-124); u is a float or a vector of embedding_dim(l) floats."""
    if isinstance(D, CountDomain):
        D = OrdinalDomain(0, D.max_count)
    if l.embedding_dim > 1:
        u = np.asarray(u, dtype=float)
        if isinstance(D, CategoricalDomain) and isinstance(l, (_l.MultinomialLoss, _l.OvALoss)):
            return int(np.argmax(u)) + 1
        if isinstance(D, OrdinalDomain) and isinstance(l, _l.OrdisticLoss):
            return int(np.argmin(u ** 2)) + 1
        if isinstance(D, OrdinalDomain) and isinstance(l, _l.MultinomialOrdinalLoss):
            eu = np.exp(_l.enforce_MNLOrdRules(u))
            p = np.concatenate([[1 - eu[0]], -np.diff(eu), [eu[-1]]])
            return int(np.argmax(p)) + 1
        if isinstance(D, OrdinalDomain):
            try:
                vals = [l.evaluate(u, i) for i in range(D.min, D.max + 1)]
            except IndexError:  # BoundsError in the reference (MultinomialLoss indexes u[a])
                raise TypeError(f"levels of {D!r} exceed {l!r}")
            if isinstance(l, _l.MultinomialLoss) and D.min < 1:
                raise TypeError(f"levels of {D!r} exceed {l!r}")
            return D.min + int(np.argmin(vals))
        raise TypeError(f"no impute method for ({D!r}, {l!r}, Vector)")
    u = float(u)
    if isinstance(D, (RealDomain, PeriodicDomain)):
        if isinstance(l, _DIFF):
            return u
        if isinstance(l, _l.PoissonLoss):
            return _l._exp(u)
        if isinstance(l, _l.OrdinalHingeLoss):
            return roundcutoff(u, l.min, l.max)
        if isinstance(l, _l.WeightedHingeLoss):
            return 1 / u if u != 0 else math.copysign(math.inf, u)
        raise ValueError("Logistic loss always imputes either +inf or -inf given a in R")
    if isinstance(D, BoolDomain):
        if l.classification:
            return u >= 0
        return not (l.evaluate(u, False) < l.evaluate(u, True))
    if isinstance(D, OrdinalDomain):
        if isinstance(l, _DIFF) or isinstance(l, _l.OrdinalHingeLoss):
            return roundcutoff(u, D.min, D.max)
        if isinstance(l, _l.PoissonLoss):
            return roundcutoff(_l._exp(u), D.min, D.max)
        if isinstance(l, _l.LogisticLoss):
            return D.max if u > 0 else D.min
        if isinstance(l, _l.WeightedHingeLoss):
            inv = 1 / u if u != 0 else math.copysign(math.inf, u)
            return roundcutoff(math.ceil(inv) if u > 0 else math.floor(inv), D.min, D.max) if math.isfinite(inv) else (D.max if inv > 0 else D.min)
    raise TypeError(f"no impute method for ({D!r}, {l!r})")


def pos_mod(T, x):
    return math.fmod(x, T) if x > 0 else math.fmod(x, T) + T


def error_metric_entry(D, l, u, a):
    """error_metric(D, l, u, a): squared error or 0-1 misclassification of the imputed value (src/impute_and_err.jl:48-130)."""
    imp = impute_entry(D, l, u)
    if isinstance(D, (BoolDomain, CategoricalDomain)):
        return float(not (float(imp) == float(a)))
    if isinstance(D, PeriodicDomain):
        return (pos_mod(D.T, float(imp)) - pos_mod(D.T, float(a))) ** 2
    return (float(imp) - float(a)) ** 2


def _resolve(glrm, domains):
    return [default_domain(l) for l in glrm.losses] if domains is None else list(domains)


def impute(glrm, X=None, Y=None, domains=None, *, engine=None):
    """impute(glrm) = impute(domains, losses, X'Y): the full m x n matrix of imputed values (Bool columns as 1.0 / 0.0)."""
    from .fit import _ensure_handle
    from .params import HipProxGradParams
    api = engine if engine is not None else _capi.hip_api()
    X = np.asfortranarray(glrm.X if X is None else X, dtype=np.float64)
    Y = np.asfortranarray(glrm.Y if Y is None else Y, dtype=np.float64)
    h = _ensure_handle(glrm, api, HipProxGradParams(), allow_dense=False)[0]
    return api.impute(h, X, Y, pack_domains(_resolve(glrm, domains)), glrm.m, glrm.n)


def impute_missing(glrm, **kw):
    """impute(glrm) with the observed entries copied back (src/evaluate_fit.jl:157-165)."""
    Ahat = impute(glrm, **kw)
    J = np.repeat(np.arange(glrm.n), np.diff(glrm._colptr))
    Ahat[glrm._rowidx, J] = glrm._colvals
    return Ahat


def error_metric(glrm, X=None, Y=None, domains=None, *, standardize=False, engine=None):
    """error_metric(glrm, X, Y, domains; standardize) over observed_examples (src/evaluate_fit.jl:107-153)."""
    from .fit import _ensure_handle
    from .params import HipProxGradParams
    api = engine if engine is not None else _capi.hip_api()
    X = np.asfortranarray(glrm.X if X is None else X, dtype=np.float64)
    Y = np.asfortranarray(glrm.Y if Y is None else Y, dtype=np.float64)
    h = _ensure_handle(glrm, api, HipProxGradParams(), allow_dense=False)[0]
    return api.error_metric(h, X, Y, pack_domains(_resolve(glrm, domains)), standardize)
