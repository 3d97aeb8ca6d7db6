"""ctypes binding of the C ABI declared in include/glrm_hip.h.

The product binds ``libglrm_hip.so`` with the ``glrm_hip_`` prefix (:func:`hip_api`).  The
same binder is reused by tests/ to bind the CPU oracle (``glrm_cpu_`` prefix) -- the product
never does that, and :func:`hip_api` raises if the HIP library is missing (no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ABI_VERSION = 3

GLRM_OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_COMM, ERR_OOM, ERR_NONFINITE = -1, -2, -3, -4, -5, -6
PROBLEM_DEVICE_ARRAYS = 1
PROBLEM_DEFER_SETUP = 2
PROBLEM_BORROW_DEVICE_ARRAYS = 4
PROBLEM_ROWS_FROM_COLS = 8   # Omega is a sparse matrix's pattern: hand over the column view only, the engine derives the row view


class GLRMError(RuntimeError):
    """A negative glrm_status from the engine (the reference throws Julia exceptions instead)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[glrm status {code}] {message}")
        self.code = code
        self.message = message


class CLoss(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dim", C.c_int32), ("scale", C.c_double), ("p0", C.c_double), ("p1", C.c_double)]


class CReg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("wrap", C.c_int32), ("scale", C.c_double)]


LOSS_DTYPE = np.dtype([("kind", "<i4"), ("dim", "<i4"), ("scale", "<f8"), ("p0", "<f8"), ("p1", "<f8")])
REG_DTYPE = np.dtype([("kind", "<i4"), ("wrap", "<i4"), ("scale", "<f8")])
DOMAIN_DTYPE = np.dtype([("kind", "<i4"), ("reserved", "<i4"), ("lo", "<f8"), ("hi", "<f8")])  # glrm_domain, 24 bytes
assert LOSS_DTYPE.itemsize == C.sizeof(CLoss) == 32 and REG_DTYPE.itemsize == C.sizeof(CReg) == 16


class CProblem(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64), ("k", C.c_int32), ("flags", C.c_int32),
        ("row_begin", C.c_int64), ("row_end", C.c_int64), ("col_begin", C.c_int64), ("col_end", C.c_int64),
        ("rowptr", C.c_void_p), ("colidx", C.c_void_p), ("rowvals", C.c_void_p),
        ("colptr", C.c_void_p), ("rowidx", C.c_void_p), ("colvals", C.c_void_p),
        ("losses", C.c_void_p), ("n_losses", C.c_int64),
        ("rx", C.c_void_p), ("n_rx", C.c_int64),
        ("ry", C.c_void_p), ("n_ry", C.c_int64),
        ("dense_A", C.c_void_p), ("dense_ld", C.c_int64), ("dense_colmajor", C.c_int32), ("dense_reserved", C.c_int32),
    ]


class CParams(C.Structure):
    _fields_ = [
        ("stepsize", C.c_double), ("max_iter", C.c_int64), ("inner_iter_X", C.c_int64), ("inner_iter_Y", C.c_int64),
        ("abs_tol", C.c_double), ("rel_tol", C.c_double), ("min_stepsize", C.c_double),
    ]


class CSparseParams(C.Structure):
    _fields_ = [("stepsize", C.c_double), ("max_iter", C.c_int64), ("inner_iter", C.c_int64), ("abs_tol", C.c_double),
                ("min_stepsize", C.c_double)]


class COptions(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("profile", C.c_int32), ("waves_row", C.c_int32), ("waves_col", C.c_int32),
                ("stream", C.c_void_p), ("caller_stream", C.c_int32), ("tiled", C.c_int32), ("quad_gram", C.c_int32),
                ("sum_order", C.c_int32), ("reserved0", C.c_int32), ("reserved", C.c_int32)]


assert C.sizeof(COptions) == 48


class CArrival(C.Structure):
    """glrm_arrival: rows [begin, end) of X are complete on the device once ``event`` (a hipEvent_t, 0 = already there) has fired."""
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("event", C.c_void_p)]


class CSignature(C.Structure):
    """glrm_signature: what the kernel choice reads from the WHOLE problem (sum the counts over shards, max the rest)."""
    _fields_ = [("nnz_rows", C.c_int64), ("nnz_cols", C.c_int64), ("max_row_len", C.c_int64), ("max_col_len", C.c_int64),
                ("rows_unordered", C.c_int32), ("cols_unordered", C.c_int32)]
    SUM_FIELDS = ("nnz_rows", "nnz_cols")

    def astuple(self):
        return tuple(int(getattr(self, f)) for f, _ in self._fields_)

    @classmethod
    def combine(cls, parts):
        """The whole problem's signature from the shards' (each a CSignature or its astuple())."""
        parts = [p.astuple() if isinstance(p, cls) else tuple(int(v) for v in p) for p in parts]
        out = cls()
        for i, (f, _) in enumerate(cls._fields_):
            vals = [p[i] for p in parts]
            setattr(out, f, sum(vals) if f in cls.SUM_FIELDS else max(vals))
        return out


class CMultiOptions(C.Structure):
    _fields_ = [("n_shards", C.c_int32), ("exchange", C.c_int32), ("device_ids", C.c_void_p), ("x_chunks", C.c_int32), ("arrival", C.c_int32)]


class CKernelStats(C.Structure):
    _fields_ = [
        ("launches_x", C.c_int64), ("launches_y", C.c_int64), ("ms_x", C.c_double), ("ms_y", C.c_double),
        ("trials_x", C.c_int64), ("trials_y", C.c_int64), ("accepts_x", C.c_int64), ("accepts_y", C.c_int64),
        ("nnz_rows", C.c_int64), ("nnz_cols", C.c_int64),
        ("waves_row", C.c_int32), ("waves_col", C.c_int32), ("ld", C.c_int32), ("tiled", C.c_int32),
        ("ms_wait_y", C.c_double),
    ]

    def asdict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class CSumOrder(C.Structure):
    """glrm_sum_order: the order in which a handle adds the terms of a segment's sums (include/glrm_hip.h)."""
    _fields_ = [("family", C.c_int32), ("lanes", C.c_int32), ("comps", C.c_int32), ("waves", C.c_int32),
                ("waves4_from", C.c_int64), ("waves8_from", C.c_int64), ("cached_maxlen", C.c_int64),
                ("cached_waves", C.c_int32), ("batch", C.c_int32), ("batch_one_wave_only", C.c_int32), ("rotate", C.c_int32),
                ("window", C.c_int64), ("windows_per_sup", C.c_int64), ("private_order", C.c_int32), ("long_from", C.c_int32)]
    FAMILIES = {0: "reference", 1: "strided", 2: "windowed", 3: "other"}

    def asdict(self):
        d = {f: int(getattr(self, f)) for f, _ in self._fields_}
        d["family_name"] = self.FAMILIES.get(d["family"], "?")
        return d


assert C.sizeof(CSumOrder) == 80

#: every symbol include/glrm_hip.h declares (suffix after the prefix); the CPU test-suite checks
#: that the built library exports all of them.
ABI_SYMBOLS = (
    "version", "last_error", "create", "destroy", "signature", "finalize", "fit", "fit_sparse", "objective", "factor_ld", "bind_buffers",
    "set_factors", "get_factors", "reset_stepsizes", "step_x", "step_y", "step_y_arrival", "step_x_range", "gradstep_x", "gradstep_y", "col_losses", "row_penalties",
    "col_penalties", "set_regularizers", "subset", "init_svd", "error_metric", "impute", "sum", "synchronize", "kernel_stats", "sum_order",
    "multi_create", "multi_fit", "multi_set_regularizers", "multi_info", "multi_destroy",
)


#: Bumped whenever a loss / regularizer object is created or modified or a model's descriptor list changes: lets a model reuse
#: its packed descriptors (and its engine handle) without re-reading a million Python objects per fit! call.
EPOCH = [0]


def bump_epoch():
    EPOCH[0] += 1


class TrackedList(list):
    """A list whose mutations bump EPOCH (the losses / rx / ry lists of a GLRM)."""

    def _mut(name):  # noqa: N805
        base = getattr(list, name)

        def f(self, *a, **k):
            bump_epoch()
            return base(self, *a, **k)
        f.__name__ = name
        return f

    for _n in ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove", "clear", "sort", "reverse"):
        locals()[_n] = _mut(_n)
    del _n, _mut


def _ptr(a):
    """numpy array -> address; int -> address; None -> NULL."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return int(a)
    return a.ctypes.data


class Api:
    """One loaded library + symbol prefix.  Thin, stateless, no numerics."""

    def __init__(self, lib: C.CDLL, prefix: str, device_type: str):
        self.lib, self.prefix, self.device_type = lib, prefix, device_type
        self.dense_ok = prefix == "glrm_hip_"  # the dense (matrix-core) hand-over exists in the HIP engine only
        H = C.c_void_p
        sig = {
            "version": (C.c_int, []),
            "last_error": (C.c_char_p, []),
            "create": (C.c_int, [C.POINTER(H), C.POINTER(CProblem), C.POINTER(COptions)]),
            "destroy": (None, [H]),
            "signature": (C.c_int, [H, C.POINTER(CSignature)]),
            "finalize": (C.c_int, [H, C.POINTER(CSignature)]),
            "fit": (C.c_int, [H, C.POINTER(CParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                              C.POINTER(C.c_int64)]),
            "fit_sparse": (C.c_int, [H, C.POINTER(CSparseParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.POINTER(C.c_int64)]),
            "gradstep_x": (C.c_int, [H, C.c_double]),
            "gradstep_y": (C.c_int, [H, C.c_double]),
            "objective": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
            "factor_ld": (C.c_int, [H]),
            "bind_buffers": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
            "set_factors": (C.c_int, [H, C.c_void_p, C.c_void_p]),
            "get_factors": (C.c_int, [H, C.c_void_p, C.c_void_p]),
            "reset_stepsizes": (C.c_int, [H, C.c_double]),
            "step_x": (C.c_int, [H, C.c_double]),
            "step_y": (C.c_int, [H, C.c_double]),
            "step_y_arrival": (C.c_int, [H, C.c_double, C.c_void_p, C.c_int32]),
            "step_x_range": (C.c_int, [H, C.c_int64, C.c_int64, C.c_double]),
            "col_losses": (C.c_int, [H]),
            "row_penalties": (C.c_int, [H]),
            "col_penalties": (C.c_int, [H]),
            "set_regularizers": (C.c_int, [H, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
            "subset": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(H)]),
            "init_svd": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_uint64, C.c_void_p, C.POINTER(C.c_int32)]),
            "error_metric": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
            "impute": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
            "sum": (C.c_int, [H, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]),
            "synchronize": (C.c_int, [H]),
            "kernel_stats": (C.c_int, [H, C.POINTER(CKernelStats), C.c_int]),
            "sum_order": (C.c_int, [H, C.c_int32, C.POINTER(CSumOrder)]),
            "multi_create": (C.c_int, [C.POINTER(H), C.POINTER(CProblem), C.POINTER(COptions), C.POINTER(CMultiOptions)]),
            "multi_fit": (C.c_int, [H, C.POINTER(CParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.POINTER(C.c_int64)]),
            "multi_set_regularizers": (C.c_int, [H, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
            "multi_info": (C.c_int, [H, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
            "multi_destroy": (None, [H]),
        }
        self._f = {}
        for name, (res, args) in sig.items():
            fn = getattr(lib, prefix + name)
            fn.restype, fn.argtypes = res, args
            self._f[name] = fn
        v = self._f["version"]()
        if v != ABI_VERSION:
            raise RuntimeError(f"{prefix}version() = {v}, this host layer speaks ABI {ABI_VERSION}")

    # -- error plumbing ---------------------------------------------------------------
    def last_error(self) -> str:
        m = self._f["last_error"]()
        return m.decode("utf-8", "replace") if m else ""

    def _ck(self, rc: int):
        if rc != GLRM_OK:
            raise GLRMError(rc, self.last_error())

    # -- lifecycle --------------------------------------------------------------------
    @staticmethod
    def _cproblem(prob):
        p = CProblem()
        p.m, p.n, p.k, p.flags = prob.m, prob.n, prob.k, prob.flags
        p.row_begin, p.row_end, p.col_begin, p.col_end = prob.row_begin, prob.row_end, prob.col_begin, prob.col_end
        p.rowptr, p.colidx, p.rowvals = _ptr(prob.rowptr), _ptr(prob.colidx), _ptr(prob.rowvals)
        p.colptr, p.rowidx, p.colvals = _ptr(prob.colptr), _ptr(prob.rowidx), _ptr(prob.colvals)
        p.losses, p.n_losses = _ptr(prob.losses), len(prob.losses)
        p.rx, p.n_rx = _ptr(prob.rx), len(prob.rx)
        p.ry, p.n_ry = _ptr(prob.ry), len(prob.ry)
        if prob.dense_A is not None:
            p.dense_A, p.dense_ld, p.dense_colmajor, p.dense_reserved = _ptr(prob.dense_A), prob.dense_ld, prob.dense_colmajor, 0
        return p

    def create(self, prob: "ProblemArrays", device_id=-1, profile=0, waves_row=0, waves_col=0, stream=None, tiled=0, quad_gram=0, defer=False,
               sum_order=0):
        """``stream=None``: the handle creates a private stream.  ``stream=<int>``: launch on exactly that
        hipStream_t -- 0 is the legacy default stream (what torch.cuda.current_stream().cuda_stream returns
        for the default stream), so kernels stay ordered with the caller's other work on it.
        ``defer=True`` (one shard of a sharded fit): only upload; the host combines :meth:`signature` over the shards and
        calls :meth:`finalize` on every one of them (GLRM_PROBLEM_DEFER_SETUP).
        ``sum_order=1``: the reference-order validation sweeps (glrm_options.sum_order)."""
        if prob.dense_A is not None and not self.dense_ok:
            raise GLRMError(ERR_UNSUPPORTED, "this engine takes observation lists only")
        p = self._cproblem(prob)
        if defer:
            p.flags |= PROBLEM_DEFER_SETUP
        o = COptions(device_id, profile, waves_row, waves_col, (stream or None), 0 if stream is None else 1, tiled, int(quad_gram), int(sum_order),
                     0, 0)
        h = C.c_void_p()
        self._ck(self._f["create"](C.byref(h), C.byref(p), C.byref(o)))
        return h

    def signature(self, h) -> CSignature:
        sig = CSignature()
        self._ck(self._f["signature"](h, C.byref(sig)))
        return sig

    def finalize(self, h, whole: "CSignature | None" = None):
        self._ck(self._f["finalize"](h, C.byref(whole) if whole is not None else None))

    # -- one process, several devices (glrm_*_multi_*) -----------------------------------------
    def multi_create(self, prob: "ProblemArrays", n_shards, device_ids=None, exchange=0, x_chunks=0, profile=0, waves_row=0,
                     waves_col=0, tiled=0, quad_gram=0, arrival=0, sum_order=0):
        """The whole problem (host arrays), sharded by the library over ``device_ids`` (default 0..n_shards-1; ids may repeat)."""
        if prob.dense_A is not None and not self.dense_ok:
            raise GLRMError(ERR_UNSUPPORTED, "this engine takes observation lists only")
        p = self._cproblem(prob)
        o = COptions(-1, profile, waves_row, waves_col, None, 0, tiled, int(quad_gram), int(sum_order), 0, 0)
        ids = None if device_ids is None else np.ascontiguousarray(device_ids, dtype=np.int32)
        if ids is not None and len(ids) != n_shards:
            raise ValueError("device_ids must have n_shards entries")
        mo = CMultiOptions(int(n_shards), int(exchange), _ptr(ids), int(x_chunks), int(arrival))
        h = C.c_void_p()
        self._ck(self._f["multi_create"](C.byref(h), C.byref(p), C.byref(o), C.byref(mo)))
        return h

    def multi_fit(self, mh, params, X, Y):
        cap = int(params.max_iter) + 1
        obj, sec = np.zeros(cap), np.zeros(cap)
        nrec = C.c_int64(0)
        cp = CParams(params.stepsize, params.max_iter, params.inner_iter_X, params.inner_iter_Y, params.abs_tol,
                     params.rel_tol, params.min_stepsize)
        self._ck(self._f["multi_fit"](mh, C.byref(cp), _ptr(X), _ptr(Y), _ptr(obj), _ptr(sec), cap, C.byref(nrec)))
        return obj[: nrec.value].copy(), sec[: nrec.value].copy()

    def multi_set_regularizers(self, mh, rx, ry):
        self._ck(self._f["multi_set_regularizers"](mh, _ptr(rx), len(rx), _ptr(ry), len(ry)))

    def multi_info(self, mh, n_shards):
        rb, cb = np.zeros(n_shards + 1, np.int64), np.zeros(n_shards + 1, np.int64)
        ex, ms = C.c_int32(0), C.c_double(0.0)
        self._ck(self._f["multi_info"](mh, _ptr(rb), _ptr(cb), C.byref(ex), C.byref(ms)))
        return {"row_bounds": rb.tolist(), "col_bounds": cb.tolist(), "exchange": ex.value, "exchange_ms": ms.value}

    def multi_destroy(self, mh):
        if mh is not None and mh.value:
            self._f["multi_destroy"](mh)
            mh.value = None

    def destroy(self, h):
        if h is not None and h.value:
            self._f["destroy"](h)
            h.value = None

    # -- whole-fit --------------------------------------------------------------------
    def fit(self, h, params, X, Y):
        cap = int(params.max_iter) + 1
        obj = np.zeros(cap)
        sec = np.zeros(cap)
        nrec = C.c_int64(0)
        cp = CParams(params.stepsize, params.max_iter, params.inner_iter_X, params.inner_iter_Y, params.abs_tol,
                     params.rel_tol, params.min_stepsize)
        self._ck(self._f["fit"](h, C.byref(cp), _ptr(X), _ptr(Y), _ptr(obj), _ptr(sec), cap, C.byref(nrec)))
        return obj[: nrec.value].copy(), sec[: nrec.value].copy()

    def fit_sparse(self, h, params, X, Y):
        cap = int(params.max_iter) + 2
        obj, sec = np.zeros(cap), np.zeros(cap)
        nrec = C.c_int64(0)
        cp = CSparseParams(params.stepsize, params.max_iter, params.inner_iter, params.abs_tol, params.min_stepsize)
        self._ck(self._f["fit_sparse"](h, C.byref(cp), _ptr(X), _ptr(Y), _ptr(obj), _ptr(sec), cap, C.byref(nrec)))
        return obj[: nrec.value].copy(), sec[: nrec.value].copy()

    def gradstep_x(self, h, alpha):
        self._ck(self._f["gradstep_x"](h, float(alpha)))

    def gradstep_y(self, h, alpha):
        self._ck(self._f["gradstep_y"](h, float(alpha)))

    def objective(self, h, X, Y, include_reg=True) -> float:
        out = C.c_double(0.0)
        self._ck(self._f["objective"](h, _ptr(X), _ptr(Y), 1 if include_reg else 0, C.byref(out)))
        return out.value

    # -- step-level -------------------------------------------------------------------
    def factor_ld(self, h) -> int:
        ld = self._f["factor_ld"](h)
        if ld <= 0:
            raise GLRMError(ld, self.last_error())
        return ld

    def bind_buffers(self, h, dX, dY, dObjCol, dObjRow):
        self._ck(self._f["bind_buffers"](h, _ptr(dX), _ptr(dY), _ptr(dObjCol), _ptr(dObjRow)))

    def set_factors(self, h, X, Y):
        self._ck(self._f["set_factors"](h, _ptr(X), _ptr(Y)))

    def get_factors(self, h, X, Y):
        self._ck(self._f["get_factors"](h, _ptr(X), _ptr(Y)))

    def reset_stepsizes(self, h, stepsize):
        self._ck(self._f["reset_stepsizes"](h, float(stepsize)))

    def step_x(self, h, min_stepsize):
        self._ck(self._f["step_x"](h, float(min_stepsize)))

    def step_x_range(self, h, seg_begin, seg_end, min_stepsize):
        self._ck(self._f["step_x_range"](h, int(seg_begin), int(seg_end), float(min_stepsize)))

    def step_y(self, h, min_stepsize):
        self._ck(self._f["step_y"](h, float(min_stepsize)))

    def step_y_arrival(self, h, min_stepsize, blocks):
        """The Y half-step while X is arriving: ``blocks`` = [(begin, end, event)] tiling the rows [0, m) in the order the host
        expects them; ``event`` is a hipEvent_t as an integer (torch.cuda.Event(...).cuda_event) or 0 / None for rows already there."""
        arr = (CArrival * max(len(blocks), 1))()
        for i, (b, e, ev) in enumerate(blocks):
            arr[i] = CArrival(int(b), int(e), int(ev) if ev else None)
        self._ck(self._f["step_y_arrival"](h, float(min_stepsize), C.cast(arr, C.c_void_p), len(blocks)))

    def col_losses(self, h):
        self._ck(self._f["col_losses"](h))

    def row_penalties(self, h):
        self._ck(self._f["row_penalties"](h))

    def col_penalties(self, h):
        self._ck(self._f["col_penalties"](h))

    def set_regularizers(self, h, rx, ry):
        """rx, ry: REG_DTYPE arrays with the same lengths as at create."""
        self._ck(self._f["set_regularizers"](h, _ptr(rx), len(rx), _ptr(ry), len(ry)))

    def subset(self, h, row_tags, col_tags, match, invert=False):
        """Child handle over the entries whose tag (uint8 per entry of the parent's row view / column view) equals
        ``match`` (or differs from it when ``invert``): the train / test split of the cross-validation drivers, compacted
        from the parent's resident data.  The child of a SHARD parent (row / column ranges not the whole problem) comes back
        deferred: read ``signature`` of every shard's child, combine (sum the counts, max the rest) and ``finalize`` each child
        with the whole subset problem's signature before the first step-level call (include/glrm_hip.h, glrm_hip_subset)."""
        row_tags = np.ascontiguousarray(row_tags, dtype=np.uint8)
        col_tags = np.ascontiguousarray(col_tags, dtype=np.uint8)
        out = C.c_void_p()
        self._ck(self._f["subset"](h, _ptr(row_tags), _ptr(col_tags), int(match), 1 if invert else 0, C.byref(out)))
        return out

    def init_svd(self, h, X, Y, max_iter=0, tol=0.0, seed=1):
        """init_svd! on the handle's resident lists: fills X (k x m) and Y (k x d); returns (singular values, iterations)."""
        sv = np.zeros(X.shape[0])
        it = C.c_int32(0)
        self._ck(self._f["init_svd"](h, _ptr(X), _ptr(Y), int(max_iter), float(tol), int(seed), _ptr(sv), C.byref(it)))
        return sv, it.value

    def error_metric(self, h, X, Y, domains, standardize=False) -> float:
        """domains: DOMAIN_DTYPE array, one per column."""
        out = C.c_double(0.0)
        self._ck(self._f["error_metric"](h, _ptr(X), _ptr(Y), _ptr(domains), 1 if standardize else 0, C.byref(out)))
        return out.value

    def impute(self, h, X, Y, domains, m, n):
        Ahat = np.zeros((m, n), order="F")
        self._ck(self._f["impute"](h, _ptr(X), _ptr(Y), _ptr(domains), _ptr(Ahat)))
        return Ahat

    def sum(self, h, vec, n) -> float:
        out = C.c_double(0.0)
        self._ck(self._f["sum"](h, _ptr(vec), int(n), C.byref(out)))
        return out.value

    def synchronize(self, h):
        self._ck(self._f["synchronize"](h))

    def kernel_stats(self, h, reset=False) -> dict:
        st = CKernelStats()
        self._ck(self._f["kernel_stats"](h, C.byref(st), 1 if reset else 0))
        return st.asdict()


    def sum_order(self, h, which) -> CSumOrder:
        """which: 0 = row view (X half-step), 1 = column view (Y half-step)."""
        o = CSumOrder()
        self._ck(self._f["sum_order"](h, int(which), C.byref(o)))
        return o


class ProblemArrays:
    """Plain container for one shard in the ABI's layout (0-based, CSR + CSC, descriptors)."""

    def __init__(self, m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, losses, rx, ry,
                 row_begin=0, row_end=None, col_begin=0, col_end=None, flags=0, dense_A=None, dense_ld=0, dense_colmajor=0):
        self.m, self.n, self.k, self.flags = int(m), int(n), int(k), int(flags)
        self.row_begin, self.row_end = int(row_begin), int(m if row_end is None else row_end)
        self.col_begin, self.col_end = int(col_begin), int(n if col_end is None else col_end)
        self.rowptr, self.colidx, self.rowvals = rowptr, colidx, rowvals
        self.colptr, self.rowidx, self.colvals = colptr, rowidx, colvals
        self.losses, self.rx, self.ry = losses, rx, ry  # numpy structured arrays (LOSS_DTYPE / REG_DTYPE)
        # fully observed QuadLoss hand-over: the whole m x n matrix (numpy array or device address) instead of lists
        self.dense_A, self.dense_ld, self.dense_colmajor = dense_A, int(dense_ld), int(dense_colmajor)

    @property
    def ystart(self):
        """Column f owns vectors [ystart[f], ystart[f+1]) of Y (get_yidxs, src/losses.jl:76-93)."""
        dims = np.maximum(np.asarray(self.losses["dim"], dtype=np.int64), 1)
        if len(dims) == 1:
            return np.arange(self.n + 1, dtype=np.int64) * int(dims[0])
        return np.concatenate([[0], np.cumsum(dims)]).astype(np.int64)

    @property
    def d(self):
        """Vectors of Y = sum of the embedding dimensions (= n for scalar losses)."""
        return int(self.ystart[-1])


_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# GLRM_HIP_LIB_PATH: another BUILD of the same engine (tests/perf/ab_lib.py, variant builds of build.py) -- never another engine
HIP_LIB_PATH = os.environ.get("GLRM_HIP_LIB_PATH") or os.path.join(_PKG_DIR, "libglrm_hip.so")
_hip_api = None


HIP_TESTING_LIB_PATH = os.path.join(_PKG_DIR, "libglrm_hip_testing.so")
_hip_testing_api = None


def hip_testing_api() -> Api:
    """The TEST BUILD of the engine (libglrm_hip_testing.so: the product objects with csrc/glrm_testhooks.hip and csrc/glrm_multigpu.hip rebuilt
    under -DGLRM_HIP_TESTING): the only library in which GLRM_HIP_TEST_FAIL_FINALIZE, GLRM_HIP_RCCL_LIB / GLRM_HIP_RCCL_ALLOW_SHARED and the
    link emulator (GLRM_EXCHANGE_EMULATE_*) exist.  For tests and `bench.py --emulate-link-gbps`; never what `fit_b` runs on."""
    global _hip_testing_api
    if _hip_testing_api is None:
        if not os.path.exists(HIP_TESTING_LIB_PATH):
            raise RuntimeError(f"{HIP_TESTING_LIB_PATH} is missing: build it with `python __graft_entry__.py`")
        _hip_testing_api = Api(C.CDLL(HIP_TESTING_LIB_PATH, mode=C.RTLD_LOCAL), "glrm_hip_", "cuda")
    return _hip_testing_api


def hip_api() -> Api:
    """The MI355X engine.  Fails loudly when the HIP library has not been built -- there is no
    CPU fallback in the product path."""
    global _hip_api
    if _hip_api is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(
                f"{HIP_LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
                "lowrankmodels.jl_amd has no CPU fallback."
            )
        _hip_api = Api(C.CDLL(HIP_LIB_PATH, mode=C.RTLD_GLOBAL), "glrm_hip_", "cuda")
    return _hip_api
