"""cross_validate / cv_by_iter / regularization_path / get_train_and_test (reference: src/cross_validate.jl:1-240) on the
MI355X engine, with the driver-level fusion of SURVEY.md section 8(f) rank 2: the data matrix is uploaded once; every
train / test model of a fold is a ``glrm_hip_subset`` of the resident parent handle (one tag byte per observation crosses
PCIe instead of 12 bytes per observation and view), the regularization path re-fits on one handle through
``glrm_hip_set_regularizers`` and warm starts, and held-out errors are evaluated by the device objective.

Random choices (fold labels, hold-out draws) come from ``rng`` (numpy Generator) or are passed in as ``groups`` -- Julia's
``rand!`` stream cannot be reproduced, the partition logic given the labels is the reference's.
"""
from __future__ import annotations

import copy as _copy

import numpy as np

from . import _capi
from .convergence import ConvergenceHistory
from .fit import _ensure_handle, fit_b, objective
from .glrm import copy_estimate, parameter_estimate, scale_regularizer_
from .params import Params, ProxGradParams


def loss_fn(glrm, X=None, Y=None, **kw):
    """The default error metric: the objective minus the regularization (src/cross_validate.jl:3-5)."""
    return objective(glrm, X, Y, include_regularization=False, **kw)


def flatten_observations(observed_features):
    """[(i, j) for i, row in enumerate(observed_features) for j in row] as two index arrays (src/cross_validate.jl:107-115)."""
    lens = np.array([len(r) for r in observed_features], dtype=np.int64)
    I = np.repeat(np.arange(len(observed_features), dtype=np.int64), lens)
    J = np.concatenate([np.asarray(r, dtype=np.int64) for r in observed_features]) if lens.sum() else np.zeros(0, dtype=np.int64)
    return I, J


def _lists(ptr, idx):
    return [idx[ptr[s]:ptr[s + 1]] for s in range(len(ptr) - 1)]


def _stable_order_by_column(rowptr, colidx, n):
    """Positions of the row-view entries listed column by column, obs order kept inside a column (what sort_observations'
    push! produces).  CSR -> CSC transposition of the entry numbers when scipy is there (linear time), a stable argsort
    otherwise."""
    nnz = len(colidx)
    try:
        import scipy.sparse as sp
        if nnz < 2 ** 31:
            csr = sp.csr_matrix((np.arange(1, nnz + 1, dtype=np.int32 if nnz < 2 ** 31 - 1 else np.int64), colidx, rowptr),
                                shape=(len(rowptr) - 1, n))
            return csr.tocsc().data.astype(np.int64) - 1
    except ImportError:
        pass
    return np.argsort(colidx.astype(np.int64), kind="stable")


class _Split:
    """The row view of a model flattened in list order (= `obs`), the permutation that turns it into the column view
    sort_observations would build (stable by column, src/modify_glrm.jl:5-18), and whether the model's own column view
    is that one (then a subset of the resident column view is the fold's column view and the split can run on the device)."""

    def __init__(self, glrm):
        self.g = glrm
        self.I = np.repeat(np.arange(glrm.m, dtype=np.int64), np.diff(glrm._rowptr))
        self.J = glrm._colidx.astype(np.int64)
        self.perm = _stable_order_by_column(glrm._rowptr, glrm._colidx, glrm.n)
        counts = np.bincount(self.J, minlength=glrm.n)
        self.canon_colptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.canonical = (np.array_equal(self.canon_colptr, glrm._colptr)
                          and np.array_equal(self.I[self.perm].astype(np.int32), glrm._rowidx))

    def child(self, keep):
        """copy_estimate(glrm) with observed_features / observed_examples = sort_observations(obs[keep])."""
        g = self.g
        c = copy_estimate(g)
        keep = np.asarray(keep, dtype=bool)
        rowcount = np.bincount(self.I[keep], minlength=g.m)
        c._rowptr = np.concatenate([[0], np.cumsum(rowcount)]).astype(np.int64)
        c._colidx = np.ascontiguousarray(g._colidx[keep])
        c._rowvals = np.ascontiguousarray(g._rowvals[keep])
        kp = keep[self.perm]
        sel = self.perm[kp]
        colcount = np.bincount(self.J[sel], minlength=g.n)
        c._colptr = np.concatenate([[0], np.cumsum(colcount)]).astype(np.int64)
        c._rowidx = np.ascontiguousarray(self.I[sel].astype(np.int32))
        c._colvals = np.ascontiguousarray(g._rowvals[sel])
        c._fully_observed = False
        c._pattern_from_csc = False  # lists built here: both views go over
        return c


def _children(glrm, tags, match, params, engine, fused=True):
    """(train, test) models for `tags != match` / `tags == match`; on the fused path their engine handles are subsets of
    the parent's resident handle."""
    sp = glrm._split_cache if getattr(glrm, "_split_cache", None) is not None else _Split(glrm)
    glrm._split_cache = sp
    tags = np.ascontiguousarray(tags, dtype=np.uint8)
    train, test = sp.child(tags != match), sp.child(tags == match)
    train._split_cache = test._split_cache = None
    api = engine if engine is not None else _capi.hip_api()
    if fused and sp.canonical and "subset" in api._f:
        hp, key, soft = _ensure_handle(glrm, api, params, allow_dense=False)
        col_tags = tags[sp.perm]
        train._handle_cache = (api, api.subset(hp, tags, col_tags, match, invert=True), key, soft, False)
        test._handle_cache = (api, api.subset(hp, tags, col_tags, match, invert=False), key, soft, False)
    return train, test


def check_enough_observations(lists_ptr):
    return bool(np.all(np.diff(lists_ptr) > 0))


def getfolds(obs, nfolds, m, n, ntrials=5, do_check=True, rng=None, groups=None):
    """Partition `obs = (I, J)` into nfolds groups; returns the labels (0-based) -- the per-fold lists are produced on demand
    by the drivers (src/cross_validate.jl:54-84).  With do_check every training fold must touch every row and column."""
    I, J = obs
    rng = np.random.default_rng() if rng is None else rng
    for _ in range(ntrials):
        g = np.asarray(groups, dtype=np.int64) if groups is not None else rng.integers(0, nfolds, len(I))
        ok = True
        if do_check:
            for f in range(nfolds):
                tr = g != f
                if not (np.all(np.bincount(I[tr], minlength=m) > 0) and np.all(np.bincount(J[tr], minlength=n) > 0)):
                    ok = False
                    break
        if ok:
            return g.astype(np.uint8)
        if groups is not None:
            break
    raise ValueError("Not enough data to cross validate automatically.")


def cross_validate(glrm, nfolds=5, params=None, verbose=True, use_folds=None, error_fn=loss_fn, init=None, do_obs_check=False,
                   rng=None, groups=None, engine=None, fused=True):
    """cross_validate(glrm; nfolds, params, verbose, use_folds, error_fn, init, do_obs_check) -> (train_error, test_error,
    train_glrms, test_glrms) (src/cross_validate.jl:9-52)."""
    params = Params() if params is None else params
    use_folds = nfolds if use_folds is None else use_folds
    if nfolds > 255:
        raise ValueError("at most 255 folds")
    if verbose:
        print("flattening observations")
    obs = flatten_observations(glrm.observed_features)
    if verbose:
        print("computing CV folds")
    tags = getfolds(obs, nfolds, glrm.m, glrm.n, do_check=do_obs_check, rng=rng, groups=groups)
    train_glrms, test_glrms = [None] * nfolds, [None] * nfolds
    train_error, test_error = np.full(nfolds, np.nan), np.full(nfolds, np.nan)
    for ifold in range(use_folds):
        if verbose:
            print(f"\nforming train and test GLRM for fold {ifold + 1}")
        train, test = _children(glrm, tags, ifold, params, engine, fused)
        ntrain, ntest = int(train._rowptr[-1]), int(test._rowptr[-1])
        if verbose:
            print(f"training model on {ntrain} samples and testing on {ntest}")
            print(f"fitting train GLRM for fold {ifold + 1}")
        if init is not None:
            init(train)
        fit_b(train, params, verbose=verbose, engine=engine)
        X, Y = parameter_estimate(train)
        train_error[ifold] = error_fn(train, X, Y, engine=engine) / ntrain
        test_error[ifold] = error_fn(test, X, Y, engine=engine) / ntest
        if verbose:
            print(f"computing train and test error for fold {ifold + 1}:\n\ttrain error: {train_error[ifold]}\n\ttest error:  {test_error[ifold]}")
        train_glrms[ifold], test_glrms[ifold] = train, test
    return train_error, test_error, train_glrms, test_glrms


def get_train_and_test(glrm, holdout_proportion=.1, rng=None, groups=None, params=None, engine=None, fused=True):
    """(train_glrm, test_glrm): an observation goes to the test set if its uniform draw is < holdout_proportion
    (src/cross_validate.jl:90-105, applied to copy_estimate(glrm) like the callers do)."""
    nobs = int(glrm._rowptr[-1])
    if groups is None:
        rng = np.random.default_rng() if rng is None else rng
        groups = rng.random(nobs)
    tags = (np.asarray(groups) < holdout_proportion).astype(np.uint8)  # 1 = test
    return _children(glrm, tags, 1, Params() if params is None else params, engine, fused)


def cv_by_iter(glrm, holdout_proportion=.1, params=None, ch=None, verbose=True, rng=None, groups=None, engine=None, fused=True):
    """Train / test error after every outer iteration (src/cross_validate.jl:141-182): max_iter warm-started fits of one
    iteration each on a resident handle."""
    params = Params(100, max_iter=1, abs_tol=.01, min_stepsize=.01) if params is None else _copy.copy(params)
    ch = ConvergenceHistory("cv_by_iter") if ch is None else ch
    train, test = get_train_and_test(glrm, holdout_proportion, rng=rng, groups=groups, params=params, engine=engine, fused=fused)
    ntest = int(test._rowptr[-1])
    niters = params.max_iter
    params.max_iter = 1
    train_error, test_error = np.zeros(niters), np.zeros(niters)
    if verbose:
        print(f"{'train error':>12}{'test error':>12}")
    for it in range(niters):
        fit_b(train, params, ch=ch, verbose=False, engine=engine)
        train_error[it] = ch.objective[-1]
        test_error[it] = objective(test, *parameter_estimate(train), include_regularization=False, engine=engine) / ntest
        if verbose:
            print(f"{train_error[it]:12.4e}{test_error[it]:12.4e}")
    return train_error, test_error


def regularization_path(glrm, test_glrm=None, params=None, reg_params=None, holdout_proportion=.1, verbose=True, ch=None, rng=None,
                        groups=None, engine=None, fused=True):
    """regularization_path(glrm; ...) splits first (src/cross_validate.jl:184-209); regularization_path(train_glrm, test_glrm;
    ...) runs the path (:211-240): for every reg_param scale_regularizer!, warm-started fit!, mean train and test loss.
    Returns (train_error, test_error, train_time, reg_params)."""
    params = Params() if params is None else params
    reg_params = np.power(10.0, np.linspace(2, -2, 5)) if reg_params is None else np.asarray(reg_params, dtype=float)
    ch = ConvergenceHistory("reg_path") if ch is None else ch
    if test_glrm is None:
        if verbose:
            print("flattening observations\nsplitting train and test sets\nforming train and test GLRMs")
        train_glrm, test_glrm = get_train_and_test(glrm, holdout_proportion, rng=rng, groups=groups, params=params, engine=engine, fused=fused)
    else:
        train_glrm = glrm
    ntrain, ntest = int(train_glrm._rowptr[-1]), int(test_glrm._rowptr[-1])
    if verbose:
        print(f"training model on {ntrain} samples and testing on {ntest}")
    train_error, test_error, train_time = np.zeros(len(reg_params)), np.zeros(len(reg_params)), np.zeros(len(reg_params))
    for ip, reg_param in enumerate(reg_params):
        if verbose:
            print(f"fitting train GLRM for reg_param {reg_param}")
        scale_regularizer_(train_glrm, reg_param)
        fit_b(train_glrm, params, ch=ch, verbose=verbose, engine=engine)  # same handle: only the descriptors are replaced
        train_time[ip] = ch.times[-1]
        X, Y = parameter_estimate(train_glrm)
        train_error[ip] = objective(train_glrm, X, Y, include_regularization=False, engine=engine) / ntrain
        test_error[ip] = objective(test_glrm, X, Y, include_regularization=False, engine=engine) / ntest
        if verbose:
            print(f"computing mean train and test error for reg_param {reg_param}:\n\ttrain error: {train_error[ip]}\n\ttest error:  {test_error[ip]}")
    return train_error, test_error, train_time, reg_params
