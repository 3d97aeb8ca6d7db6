"""Device-resident synthetic GLRM workloads (measurement / test tooling).

Wraps ``libglrm_synth.so`` (csrc/glrm_synth.hip): the BASELINE-scale inputs (5e8+ observations) are
generated directly in HBM from a counter-based hash and handed to the engine as device arrays
(``GLRM_PROBLEM_DEVICE_ARRAYS``).  torch is used for device memory only.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _capi

SYNTH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libglrm_synth.so")


class SynthSpec(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("k", C.c_int32), ("q", C.c_int32), ("seed", C.c_uint64),
                ("value_model", C.c_int32), ("loss_mix", C.c_int32), ("noise", C.c_double)]


_lib = None


def synth_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SYNTH_LIB_PATH):
            raise RuntimeError(f"{SYNTH_LIB_PATH} is missing: run `python __graft_entry__.py` to build it")
        _lib = C.CDLL(SYNTH_LIB_PATH)
        V, I64 = C.c_void_p, C.c_int64
        _lib.glrm_synth_hip_rows.argtypes = [C.POINTER(SynthSpec), I64, I64, V, V, V, V]
        _lib.glrm_synth_hip_col_counts.argtypes = [C.POINTER(SynthSpec), I64, I64, V, V]
        _lib.glrm_synth_hip_cols.argtypes = [C.POINTER(SynthSpec), I64, I64, V, V, V, V]
        _lib.glrm_synth_hip_init.argtypes = [C.POINTER(SynthSpec), C.c_uint64, C.c_int, V, V, V]
        _lib.glrm_synth_hip_dense.argtypes = [C.POINTER(SynthSpec), I64, I64, V, I64, V, V, V]
        _lib.glrm_synth_hip_last_error.restype = C.c_char_p
    return _lib


def _ck(rc):
    if rc != 0:
        raise RuntimeError("glrm_synth_hip: " + (synth_lib().glrm_synth_hip_last_error() or b"invalid spec").decode())


QUAD = (0, 0, 1.0, 0.0, 0.0)
LOGISTIC = (7, 0, 1.0, 0.0, 0.0)
ORDINAL_1_5 = (6, 0, 1.0, 1.0, 5.0)


def loss_table(n, loss_mix):
    """Descriptors matching glrm_synth_colkind: all Quad, or (Quad, Logistic, OrdinalHinge(1,5))[f mod 3]."""
    if not loss_mix:
        if os.environ.get("GLRM_SYNTH_PER_COLUMN_QUAD"):  # experiment: QuadLoss everywhere, but one descriptor per column (two distinct
            # scales), so the per-observation-descriptor kernels run on data whose loss is cheap
            return np.array([(0, 0, 1.0 if f % 2 else 1.0 + 2.0 ** -40, 0.0, 0.0) for f in range(n)], dtype=_capi.LOSS_DTYPE)
        return np.array([QUAD], dtype=_capi.LOSS_DTYPE)
    kinds = [QUAD, LOGISTIC, ORDINAL_1_5]
    return np.array([kinds[f % 3] for f in range(n)], dtype=_capi.LOSS_DTYPE)


class DeviceWorkload:
    """One shard's CSR + CSC generated on the device; keeps the torch tensors alive."""

    def __init__(self, m, n, k, q, *, rows=None, cols=None, seed=20260926, value_model=0, loss_mix=0, noise=0.1,
                 rx=(1, 0, 1.0), ry=(1, 0, 1.0), device=None):
        import torch
        lib = synth_lib()
        self.torch = torch
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.spec = SynthSpec(m, n, k, q, seed, value_model, loss_mix, noise)
        self.m, self.n, self.k, self.q = m, n, k, q
        rb, re = (0, m) if rows is None else rows
        cb, ce = (0, n) if cols is None else cols
        self.rows, self.cols = (rb, re), (cb, ce)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            nzr = (re - rb) * q
            self.rowptr = torch.empty(re - rb + 1, dtype=torch.int64, device=self.device)
            self.colidx = torch.empty(max(nzr, 1), dtype=torch.int32, device=self.device)
            self.rowvals = torch.empty(max(nzr, 1), dtype=torch.float64, device=self.device)
            _ck(lib.glrm_synth_hip_rows(C.byref(self.spec), rb, re, self.rowptr.data_ptr(), self.colidx.data_ptr(),
                                        self.rowvals.data_ptr(), stream))
            counts = torch.zeros(ce - cb + 1, dtype=torch.int64, device=self.device)
            _ck(lib.glrm_synth_hip_col_counts(C.byref(self.spec), cb, ce, counts.data_ptr(), stream))
            self.colptr = torch.cumsum(counts, 0)
            nzc = int(self.colptr[-1].item())
            self.rowidx = torch.empty(max(nzc, 1), dtype=torch.int32, device=self.device)
            self.colvals = torch.empty(max(nzc, 1), dtype=torch.float64, device=self.device)
            _ck(lib.glrm_synth_hip_cols(C.byref(self.spec), cb, ce, self.colptr.data_ptr(), self.rowidx.data_ptr(),
                                        self.colvals.data_ptr(), stream))
            torch.cuda.synchronize(self.device)
        self.nnz_rows, self.nnz_cols = nzr, nzc
        self.losses = loss_table(n, loss_mix)
        self.rx = np.array([rx], dtype=_capi.REG_DTYPE)
        self.ry = np.array([ry], dtype=_capi.REG_DTYPE)

    def whole_signature(self) -> "_capi.CSignature":
        """glrm_signature of the WHOLE m x n problem this shard was cut from, computed from the generator (every row holds exactly q
        sorted observations; the column counts of all n columns are generated once more, counts only)."""
        torch = self.torch
        with torch.cuda.device(self.device):
            counts = torch.zeros(self.n + 1, dtype=torch.int64, device=self.device)
            _ck(synth_lib().glrm_synth_hip_col_counts(C.byref(self.spec), 0, self.n, counts.data_ptr(),
                                                      torch.cuda.current_stream(self.device).cuda_stream))
            longest = int(counts.max().item())
        sig = _capi.CSignature()
        sig.nnz_rows = sig.nnz_cols = self.m * self.q
        sig.max_row_len, sig.max_col_len = self.q, longest
        return sig

    def list_bytes(self) -> int:
        """Bytes of the two Omega views (index + value per observation)."""
        return (self.nnz_rows + self.nnz_cols) * 12

    def problem(self, borrow=False) -> _capi.ProblemArrays:
        """borrow=True: the engine reads these tensors in place (GLRM_PROBLEM_BORROW_DEVICE_ARRAYS) -- keep the workload alive and do
        NOT call free_sources() while the handle exists.  For problems whose lists do not fit HBM twice (C5: 120 GB)."""
        p = lambda t: int(t.data_ptr())
        flags = _capi.PROBLEM_DEVICE_ARRAYS | (_capi.PROBLEM_BORROW_DEVICE_ARRAYS if borrow else 0)
        return _capi.ProblemArrays(self.m, self.n, self.k, p(self.rowptr), p(self.colidx), p(self.rowvals), p(self.colptr),
                                   p(self.rowidx), p(self.colvals), self.losses, self.rx, self.ry, self.rows[0], self.rows[1],
                                   self.cols[0], self.cols[1], flags=flags)

    def init_factors(self, ld, init_seed=1):
        """X0 (ld x m) and Y0 (ld x n) iid N(0,1) on the device, padding rows zero."""
        torch = self.torch
        with torch.cuda.device(self.device):
            X = torch.empty(self.m * ld, dtype=torch.float64, device=self.device)
            Y = torch.empty(self.n * ld, dtype=torch.float64, device=self.device)
            _ck(synth_lib().glrm_synth_hip_init(C.byref(self.spec), init_seed, ld, X.data_ptr(), Y.data_ptr(),
                                                torch.cuda.current_stream(self.device).cuda_stream))
            torch.cuda.synchronize(self.device)
        return X, Y

    def free_sources(self):
        """Drop the generator's copies once the engine handle has made its own."""
        self.rowptr = self.colidx = self.rowvals = self.colptr = self.rowidx = self.colvals = None
        self.torch.cuda.empty_cache()


class DenseDeviceWorkload:
    """Fully observed m x n matrix generated in HBM (row-major), for the dense QuadLoss hand-over (BASELINE config 3)."""

    def __init__(self, m, n, k, *, seed=20260926, noise=0.1, rx=(0, 0, 1.0), ry=(0, 0, 1.0), device=None):
        import torch
        lib = synth_lib()
        self.torch = torch
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.spec = SynthSpec(m, n, k, n, seed, 0, 0, noise)
        self.m, self.n, self.k, self.q = m, n, k, n
        with torch.cuda.device(self.device):
            self.A = torch.empty(m * n, dtype=torch.float64, device=self.device)
            xs = torch.empty(m * k, dtype=torch.float64, device=self.device)
            ys = torch.empty(n * k, dtype=torch.float64, device=self.device)
            _ck(lib.glrm_synth_hip_dense(C.byref(self.spec), 0, m, self.A.data_ptr(), n, xs.data_ptr(), ys.data_ptr(),
                                         torch.cuda.current_stream(self.device).cuda_stream))
            torch.cuda.synchronize(self.device)
        self.nnz_rows = self.nnz_cols = m * n
        self.losses = np.array([QUAD], dtype=_capi.LOSS_DTYPE)
        self.rx = np.array([rx], dtype=_capi.REG_DTYPE)
        self.ry = np.array([ry], dtype=_capi.REG_DTYPE)

    def problem(self) -> _capi.ProblemArrays:
        return _capi.ProblemArrays(self.m, self.n, self.k, None, None, None, None, None, None, self.losses, self.rx, self.ry,
                                   flags=_capi.PROBLEM_DEVICE_ARRAYS, dense_A=int(self.A.data_ptr()), dense_ld=self.n, dense_colmajor=0)

    init_factors = DeviceWorkload.init_factors

    def free_sources(self):
        self.A = None
        self.torch.cuda.empty_cache()
