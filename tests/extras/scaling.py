"""TEST-SIDE helper (out of scope for the product, SURVEY.md section 2: `add_offset!` / `equilibrate_variance!` are constructor conveniences of
the reference, not part of the fit! path): M-estimators and the scale=true model rewrite (reference: src/losses.jl:116-352 M_estimator / avgerror,
src/modify_glrm.jl:31-82 equilibrate_variance! / prob_scale!).  Host-side pre-processing of the loss and regularizer scales;
the fit itself is unchanged.  Scalar losses only: the reference's M-estimators of the multi-dimensional losses do not run
(`a==j` on an array, adjoint row vectors passed to grad(::Vector)), so `scale=true` cannot be reproduced for them."""
from __future__ import annotations

import math

import numpy as np

import lowrankmodels.jl_amd.losses as _l


def M_estimator(l, a):
    """argmin_u sum_i l(u, a_i) in closed form (the reference's loss-specific methods)."""
    a = np.asarray(a, dtype=float)
    if isinstance(l, _l.QuadLoss):
        return float(np.mean(a))                                            # :148
    if isinstance(l, (_l.L1Loss, _l.HuberLoss, _l.OrdinalHingeLoss)):
        return float(np.median(a))                                          # :162, :179 (a heuristic), :294
    if isinstance(l, _l.QuantileLoss):
        return float(np.quantile(a, l.quantile))                            # :203 (Julia's default quantile = linear interpolation)
    if isinstance(l, _l.PeriodicLoss):                                      # :220-224
        w = 2 * math.pi * a / l.T
        return (l.T / (2 * math.pi)) * math.atan(float(np.sum(np.sin(w))) / float(np.sum(np.cos(w)))) + l.T / 2
    if isinstance(l, _l.PoissonLoss):
        return math.log(float(np.mean(a)))                                  # :243
    if isinstance(l, _l.LogisticLoss):                                      # :308-311
        d, N = float(np.sum(a != 0)), float(len(a))
        with np.errstate(divide="ignore"):
            return float(np.log(N + d) - np.log(N - d))
    if isinstance(l, _l.WeightedHingeLoss):                                 # :343-352
        npos = float(np.sum(a > 0))
        r = len(a) / npos - 1 if npos > 0 else math.inf
        return 1.0 if l.case_weight_ratio > r else (0.0 if l.case_weight_ratio == r else -1.0)
    raise NotImplementedError(f"M_estimator of {type(l).__name__} does not run in the reference either")


def avgerror(l, a):
    """(1/n) sum_i l(m, a_i) at the M-estimate m (src/losses.jl:128-132)."""
    a = np.asarray(a, dtype=float)
    m = M_estimator(l, a)
    return float(sum(l.evaluate(m, ai) for ai in a)) / len(a)


def _observed_values(glrm, i):
    return glrm._colvals[glrm._colptr[i]:glrm._colptr[i + 1]]


def equilibrate_variance_(glrm, columns_to_scale=None):
    """equilibrate_variance!(glrm): scale every column's loss by 1 / (its average loss at the M-estimate) and its Y regularizer by
    1 / var(observed values) (src/modify_glrm.jl:34-58)."""
    cols = range(glrm.n) if columns_to_scale is None else columns_to_scale
    for i in cols:
        nomissing = _observed_values(glrm, i)
        if len(nomissing) > 0:
            varlossi = avgerror(glrm.losses[i], nomissing)
            with np.errstate(all="ignore"):
                varregi = float(np.var(nomissing, ddof=1)) if len(nomissing) > 1 else math.nan
        else:
            varlossi, varregi = 1, 1
        if varlossi > 0:
            glrm.losses[i].mul_(glrm.losses[i].scale / varlossi)
        if varregi > 0:                      # NaN > 0 is false, like in Julia
            glrm.ry[i].mul_(glrm.ry[i].scale / varregi)
    glrm.close()                             # the descriptors of a cached engine handle are stale now
    return glrm
