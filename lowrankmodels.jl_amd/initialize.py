"""init_svd! (reference: src/initialize.jl:35-132) on the engine: the standardized, real-valued expansion of the observed
entries is decomposed on the device from the model's resident handle (``glrm_hip_init_svd``); glrm.X / glrm.Y are overwritten
with sqrt(S) U' and sqrt(S) V' diag(std)."""
from __future__ import annotations

import numpy as np

from . import _capi
from .fit import _ensure_handle
from .params import HipProxGradParams
from .regularizers import lastentry1


def _views_agree(glrm):
    """The expanded MATRIX of the reference has one entry per observed (e, f): the two lists must hold the same entries once."""
    m = glrm.m
    I = np.repeat(np.arange(m, dtype=np.int64), np.diff(glrm._rowptr))
    rk = I + m * glrm._colidx.astype(np.int64)
    J = np.repeat(np.arange(glrm.n, dtype=np.int64), np.diff(glrm._colptr))
    ck = glrm._rowidx.astype(np.int64) + m * J
    rk.sort()
    ck.sort()
    return len(rk) == len(ck) and (len(rk) < 2 or np.all(rk[1:] != rk[:-1])) and np.array_equal(rk, ck)


def init_svd_(glrm, offset=True, scale=True, TOL=1e-10, *, max_iter=0, tol=1e-10, seed=1, engine=None, check=True):
    """init_svd!(glrm; offset, scale, TOL) -> glrm.

    ``offset`` only takes effect in the reference when ``typeof(glrm.rx) == lastentry1`` (src/initialize.jl:37) -- glrm.rx is a
    Vector there, so the branch never runs; the same holds here (glrm.rx is a list), and ``scale`` depends on ``offset``.
    ``tol`` / ``max_iter`` / ``seed`` steer the subspace iteration that replaces Arpack's svds."""
    offset = offset and type(glrm.rx) is lastentry1  # never true, as in the reference
    scale = scale and offset
    if TOL != 1e-10:
        raise NotImplementedError("the engine uses the reference's default TOL = 1e-10 for vanishing standard deviations")
    if check and not _views_agree(glrm):
        raise ValueError("init_svd! works on the matrix of observed entries: observed_features and observed_examples must list "
                         "the same entries, each once")
    api = engine if engine is not None else _capi.hip_api()
    h = _ensure_handle(glrm, api, HipProxGradParams(), allow_dense=False)[0]
    X = np.zeros((glrm.k, glrm.m), order="F")
    Y = np.zeros((glrm.k, glrm.d), order="F")
    sv, iters = api.init_svd(h, X, Y, max_iter=max_iter, tol=tol, seed=seed)
    glrm.X[...] = X
    glrm.Y[...] = Y
    glrm._init_svd_info = dict(singular_values=sv, iterations=iters)
    return glrm
