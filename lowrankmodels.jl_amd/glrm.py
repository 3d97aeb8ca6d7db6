"""GLRM model object (reference: src/glrm.jl:9-89, src/modify_glrm.jl:5-18,
src/utilities/conveniencemethods.jl:16-49) and its lowering to the engine's CSR + CSC views.

Differences from the Julia type that a user can see:
  * indices are 0-based (Python); the Julia shim (julia/HipGLRM.jl) keeps 1-based indices;
  * the losses and regularizers of include/glrm_hip.h are accepted (nine scalar losses, five multi-dimensional ones,
    five base regularizers and their offset / ordinal wrappers); everything else is outside the accelerated path
    (SURVEY.md section 2);
  * ``scale=true`` (equilibrate_variance!, src/modify_glrm.jl:31-58) is a constructor convenience outside this build's scope (SURVEY.md
    section 2): ``scale`` takes a CALLABLE that rewrites the model's loss / regularizer scales at the point where the reference
    runs equilibrate_variance! (before add_offset!, src/glrm.jl:73-78); the replayed reference scripts pass
    tests/extras/scaling.py: equilibrate_variance_.  ``scale=True`` raises.
"""
from __future__ import annotations

import copy as _copy

import numpy as np

from ._capi import EPOCH, ProblemArrays, TrackedList
from .losses import Loss, embedding_dim, get_yidxs, pack_losses
from .regularizers import OrdinalReg, Regularizer, lastentry1, lastentry_unpenalized, pack_regs

try:  # scipy is optional: only needed for SparseMatrixCSC-like inputs
    import scipy.sparse as _sp
except Exception:  # pragma: no cover
    _sp = None


def _issparse(A):
    return _sp is not None and _sp.issparse(A)


def sort_observations(obs, m, n, check_empty=False):
    """sort_observations(obs, m, n) (src/modify_glrm.jl:5-18): split (i,j) pairs into per-row and
    per-column lists IN INPUT ORDER, duplicates kept.  Returns (rowptr, colidx, colptr, rowidx)."""
    if isinstance(obs, tuple) and len(obs) == 2 and isinstance(obs[0], np.ndarray):
        I, J = np.asarray(obs[0], dtype=np.int64), np.asarray(obs[1], dtype=np.int64)  # (I, J) index arrays
    else:
        arr = np.asarray(list(obs), dtype=np.int64).reshape(-1, 2)  # [(i1,j1), (i2,j2), ...]
        I, J = arr[:, 0], arr[:, 1]
    if I.size and (I.min() < 0 or I.max() >= m or J.min() < 0 or J.max() >= n):
        raise IndexError("observation index out of range")
    rowptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(I, minlength=m), out=rowptr[1:])
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(J, minlength=n), out=colptr[1:])
    pr = np.argsort(I, kind="stable")  # push! in obs order within each row
    pc = np.argsort(J, kind="stable")
    colidx, rowidx = J[pr].astype(np.int32), I[pc].astype(np.int32)
    if check_empty and (np.any(np.diff(rowptr) == 0) or np.any(np.diff(colptr) == 0)):
        raise ValueError("Every row and column must contain at least one observation")
    return rowptr, colidx, colptr, rowidx


def _flatten(lists, bound):
    """list of index lists -> (ptr, idx); keeps list order and duplicates."""
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    np.cumsum([len(l) for l in lists], out=ptr[1:])
    idx = np.concatenate([np.asarray(l, dtype=np.int64) for l in lists]) if ptr[-1] else np.zeros(0, np.int64)
    if idx.size and (idx.min() < 0 or idx.max() >= bound):
        raise IndexError("observation index out of range")
    return ptr, idx.astype(np.int32)


class GLRM:
    """GLRM(A, losses, rx, ry, k; X, Y, obs, observed_features, observed_examples, offset, scale,
    checknan, sparse_na).  ``losses``/``rx``/``ry`` may each be a single object (broadcast like
    conveniencemethods.jl:32-49) or a list of length n / m / n.  X is k x m, Y is k x n."""

    def __init__(self, A, losses, rx, ry, k, *, X=None, Y=None, obs=None, observed_features=None,
                 observed_examples=None, offset=False, scale=False, checknan=True, sparse_na=True, rng=None):
        if not _issparse(A):
            A = np.asarray(A) if not isinstance(A, np.ndarray) else A
            if A.ndim != 2:
                raise ValueError("A must be a matrix")
        m, n = A.shape
        k = int(k)
        # singleton -> filled copies (conveniencemethods.jl:29-49)
        losses = [_copy.copy(losses) for _ in range(n)] if isinstance(losses, Loss) else list(losses)
        rx = [_copy.copy(rx) for _ in range(m)] if isinstance(rx, Regularizer) else list(rx)
        ry = [_copy.copy(ry) for _ in range(n)] if isinstance(ry, Regularizer) else list(ry)
        # dimension checks, src/glrm.jl:38-43 (same messages)
        if len(losses) != n:
            raise ValueError("There must be as many losses as there are columns in the data matrix")
        if len(rx) != m:
            raise ValueError("There must be either one X regularizer or as many X regularizers as there are rows in the data matrix")
        if len(ry) != n:
            raise ValueError("There must be either one Y regularizer or as many Y regularizers as there are columns in the data matrix")
        for l in losses:
            if not isinstance(l, Loss) or l.kind < 0:
                raise NotImplementedError(f"{type(l).__name__} is outside the accelerated path")
        for r in list(rx) + list(ry):
            if not isinstance(r, Regularizer) or r.kind < 0:
                raise NotImplementedError(f"{type(r).__name__} is outside the accelerated path")
        rng = np.random.default_rng() if rng is None else rng
        if X is None:
            X = rng.standard_normal((k, m))  # randn(k, size(A,1)), src/glrm.jl:31
        d = embedding_dim(losses)  # Y has one column per embedding dimension (src/glrm.jl:31, src/losses.jl:72-93)
        if Y is None:
            Y = rng.standard_normal((k, d))
        X = np.asarray(X, dtype=np.float64)
        if X.shape != (k, m) and X.shape == (m, k):
            X = X.T  # "transposing X", src/glrm.jl:57-60
        if X.shape != (k, m):
            raise ValueError(f"X must be of size (k,m) where m is the number of rows in the data matrix. size(X) = {X.shape}, size(A) = {(m, n)}, k = {k}")
        Y = np.asarray(Y, dtype=np.float64)
        if Y.shape != (k, d):
            raise ValueError("Y must be of size (k,d) where d is the sum of the embedding dimensions of all the losses.")
        if scale is True:
            raise NotImplementedError("scale=true (equilibrate_variance!) is outside this build's scope (SURVEY.md section 2): pass a callable "
                                      "that rewrites the scales, e.g. tests/extras/scaling.py: equilibrate_variance_")
        if scale and d != n:
            raise NotImplementedError("scale=true with multi-dimensional losses: their M-estimators do not run in the reference either")

        self.A, self.losses, self.rx, self.ry, self.k = A, TrackedList(losses), TrackedList(rx), TrackedList(ry), k
        self._dk_cache = None
        self.X = np.array(X, dtype=np.float64, order="F")
        self.Y = np.array(Y, dtype=np.float64, order="F")
        self.m, self.n, self.d = m, n, d

        self._fully_observed = obs is None and observed_features is None and observed_examples is None and \
            not (sparse_na and _issparse(A))
        # Omega is the pattern of a sparse matrix (src/glrm.jl:46-48): both views list the same entries, each ascending -- the fit hands the
        # column view over alone and the engine derives the row view (GLRM_PROBLEM_ROWS_FROM_COLS, include/glrm_hip.h)
        self._pattern_from_csc = obs is None and sparse_na and _issparse(A)
        # observed entries, src/glrm.jl:45-55
        if obs is None and sparse_na and _issparse(A):
            csc = A.tocsc()
            csc.sort_indices()
            nz = csc.data != 0  # findall(!iszero, A): column-major order
            J = np.repeat(np.arange(n, dtype=np.int64), np.diff(csc.indptr))[nz]
            obs = (csc.indices.astype(np.int64)[nz], J)
        if obs is None:
            if observed_features is None:
                rowptr = np.arange(m + 1, dtype=np.int64) * n  # fill(1:n, m)
                colidx = np.tile(np.arange(n, dtype=np.int32), m)
            else:
                if len(observed_features) != m:
                    raise ValueError("observed_features must have one list per row")
                rowptr, colidx = _flatten(observed_features, n)
            if observed_examples is None:
                colptr = np.arange(n + 1, dtype=np.int64) * m  # fill(1:m, n)
                rowidx = np.tile(np.arange(m, dtype=np.int32), n)
            else:
                if len(observed_examples) != n:
                    raise ValueError("observed_examples must have one list per column")
                colptr, rowidx = _flatten(observed_examples, m)
        else:
            rowptr, colidx, colptr, rowidx = sort_observations(obs, m, n)
        self._rowptr, self._colidx, self._colptr, self._rowidx = rowptr, colidx, colptr, rowidx
        self._rowvals = self._gather(np.repeat(np.arange(m, dtype=np.int64), np.diff(rowptr)), colidx.astype(np.int64), checknan)
        self._colvals = self._gather(rowidx.astype(np.int64), np.repeat(np.arange(n, dtype=np.int64), np.diff(colptr)), False)
        self._handle_cache = None
        self._split_cache = None
        if scale:   # where the reference runs equilibrate_variance!(glrm): BEFORE add_offset!, src/glrm.jl:73-78
            scale(self)
        if offset:  # add_offset!(glrm), src/glrm.jl:76-78, src/modify_glrm.jl:21-24
            add_offset_(self)

    # -- values --------------------------------------------------------------------------
    def _gather(self, I, J, checknan):
        """A[i,j] for the listed entries as Float64; ClassificationLoss columns go through myBool
        (src/losses.jl:104-106): true/1 -> 1.0, false/0/-1 -> 0.0, anything else is an error."""
        A = self.A
        if _issparse(A):
            vals = np.asarray(A.tocsr()[I, J]).ravel().astype(np.float64) if I.size else np.zeros(0)
        elif A.dtype == object:
            vals = np.array([float(v) for v in A[I, J]], dtype=np.float64) if I.size else np.zeros(0)
        else:
            vals = A[I, J].astype(np.float64)
        cls = np.array([l.classification for l in self.losses], dtype=bool)
        if cls.any() and I.size:
            isc = cls[J]
            v = vals[isc]
            ok = (v == 1) | (v == 0) | (v == -1)
            if not ok.all():
                bad = np.flatnonzero(isc)[np.flatnonzero(~ok)[0]]
                raise ValueError(f"InexactError: entry ({I[bad]}, {J[bad]}) = {vals[bad]} is not a Bool label for a ClassificationLoss")
            vals[isc] = np.where(v == 1, 1.0, 0.0)
        if checknan and I.size:
            bad = np.flatnonzero(np.isnan(vals))
            if bad.size:
                raise ValueError(f"Observed value in entry ({I[bad[0]]}, {J[bad[0]]}) is NaN.")
        return vals

    # -- Omega views (reference field names) -------------------------------------------------
    @property
    def observed_features(self):
        return [self._colidx[self._rowptr[e]:self._rowptr[e + 1]] for e in range(self.m)]

    @property
    def observed_examples(self):
        return [self._rowidx[self._colptr[f]:self._colptr[f + 1]] for f in range(self.n)]

    def size(self):
        return (self.m, self.n)

    # -- lowering to the ABI -----------------------------------------------------------------
    def dense_eligible(self):
        """Fully observed, one QuadLoss for every column, rank 9..64, plain numeric matrix: the half-steps can run
        as fused GEMMs on the matrix cores (the `dense_A` hand-over of include/glrm_hip.h)."""
        return (self._fully_observed and not _issparse(self.A) and self.A.dtype != object and 8 < self.k <= 64
                and len(pack_losses(self.losses)) == 1 and self.losses[0].kind == 0
                and all(r.wrap == 0 for r in list(self.rx) + list(self.ry)))

    def problem_arrays(self, rows=None, cols=None, dense=False, cols_only=False) -> ProblemArrays:
        """cols_only (whole problem, sparse-matrix pattern): the column view alone with GLRM_PROBLEM_ROWS_FROM_COLS -- what the Julia shim
        hands over for a SparseMatrixCSC; the library derives the row view."""
        rb, re = (0, self.m) if rows is None else rows
        cb, ce = (0, self.n) if cols is None else cols
        if cols_only and not dense:
            if not getattr(self, "_pattern_from_csc", False) or rows is not None or cols is not None:
                raise ValueError("cols_only needs the whole problem of a model built from a sparse matrix's pattern")
            from ._capi import PROBLEM_ROWS_FROM_COLS
            return ProblemArrays(self.m, self.n, self.k, None, None, None, np.ascontiguousarray(self._colptr), np.ascontiguousarray(self._rowidx),
                                 np.ascontiguousarray(self._colvals), pack_losses(self.losses), pack_regs(self.rx), pack_regs(self.ry),
                                 flags=PROBLEM_ROWS_FROM_COLS)
        if dense:
            if not self.dense_eligible():
                raise ValueError("this model is not eligible for the dense hand-over")
            if getattr(self, "_dense_copy", None) is None:
                self._dense_copy = np.ascontiguousarray(self.A, dtype=np.float64)  # row-major m x n
                if np.isnan(self._dense_copy).any():
                    i, j = np.argwhere(np.isnan(self._dense_copy))[0]
                    raise ValueError(f"Observed value in entry ({i}, {j}) is NaN.")
            rx = pack_regs(self.rx[rb:re]) if re > rb else pack_regs(self.rx[:1])
            ry = pack_regs(self.ry[cb:ce]) if ce > cb else pack_regs(self.ry[:1])
            return ProblemArrays(self.m, self.n, self.k, None, None, None, None, None, None, pack_losses(self.losses), rx, ry,
                                 rb, re, cb, ce, dense_A=self._dense_copy, dense_ld=self.n, dense_colmajor=0)
        r0, r1 = self._rowptr[rb], self._rowptr[re]
        c0, c1 = self._colptr[cb], self._colptr[ce]
        losses = pack_losses(self.losses)
        rx = pack_regs(self.rx[rb:re]) if re > rb else pack_regs(self.rx[:1])
        ry = pack_regs(self.ry[cb:ce]) if ce > cb else pack_regs(self.ry[:1])
        return ProblemArrays(
            self.m, self.n, self.k,
            np.ascontiguousarray(self._rowptr[rb:re + 1] - r0), np.ascontiguousarray(self._colidx[r0:r1]),
            np.ascontiguousarray(self._rowvals[r0:r1]),
            np.ascontiguousarray(self._colptr[cb:ce + 1] - c0), np.ascontiguousarray(self._rowidx[c0:c1]),
            np.ascontiguousarray(self._colvals[c0:c1]),
            losses, rx, ry, rb, re, cb, ce)

    def _descriptor_key(self):
        """(what forces a new engine handle, what can be updated in place): losses are baked into the handle's
        validation and kernel choice; regularizer descriptors can be replaced as long as their counts stay."""
        tracked = all(isinstance(x, TrackedList) for x in (self.losses, self.rx, self.ry))
        fp = (EPOCH[0], id(self.losses), id(self.rx), id(self.ry)) if tracked else None
        if fp is not None and self._dk_cache is not None and self._dk_cache[0] == fp:
            return self._dk_cache[1]
        rx, ry = pack_regs(self.rx), pack_regs(self.ry)
        key = (pack_losses(self.losses).tobytes(), len(rx), len(ry)), (rx.tobytes(), ry.tobytes())
        self._dk_cache = (fp, key)
        return key

    def close(self):
        """Release the cached engine handle (device copies of Omega)."""
        if self._handle_cache is not None:
            api, h = self._handle_cache[:2]
            if len(self._handle_cache) > 5 and self._handle_cache[5] == "multi":  # glrm_hip_multi_* handle (HipProxGradParams(ngpus=N))
                api.multi_destroy(h)
            else:
                api.destroy(h)
            self._handle_cache = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def add_offset_(glrm):
    """add_offset!(glrm) (src/modify_glrm.jl:21-24): the last latent feature of every row is pinned to 1 and the last row
    of Y is not penalised -- an unpenalised per-column offset.  OrdinalReg / MNLOrdinalReg already exempt their last row."""
    glrm.rx = TrackedList(lastentry1(r) for r in glrm.rx)
    glrm.ry = TrackedList(lastentry_unpenalized(r) for r in glrm.ry)
    return glrm


def parameter_estimate(glrm):  # src/glrm.jl:82
    return glrm.X, glrm.Y


def scale_regularizer_(glrm, newscale):
    """scale_regularizer!(glrm, newscale), src/glrm.jl:85-89."""
    for r in list(glrm.rx) + list(glrm.ry):
        r.mul_(newscale)
    return glrm


def copy_estimate(g):  # conveniencemethods.jl:16-20: shares problem data, copies X and Y
    c = _copy.copy(g)
    c.X, c.Y = g.X.copy(order="F"), g.Y.copy(order="F")
    c._handle_cache = None
    c._split_cache = None  # the cached split captures the model it was built for (its X / Y): the copy builds its own
    return c
