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


class ZipfWorkload:
    """A power-law Omega (measurement / test tooling; SURVEY.md section 7.3 item 2, VERDICT r4 item 7: real observation patterns are
    heavy-tailed, the BASELINE recipes are exactly-q-per-row).  Row degrees and column popularities both follow a Zipf law
    w(rank) = (rank + 1)^-s (ranks assigned by a random permutation, so heavy rows / columns are scattered), with about `nnz` observations in
    total: row e draws round(nnz w_e / sum w) columns from the popularity distribution (inverse CDF), duplicates inside a row are dropped
    (a sparse matrix holds an entry once; the reference's `findall(!iszero, A)` order, src/glrm.jl:46-48: both views sorted), values come
    from the same low-rank-plus-noise model as the uniform recipes.  Pure torch, so the same generator makes the small CPU fixtures of the
    parity tests and the 1e9-observation device problems of `bench.py --degree zipf`.  Whole problem only (no shards)."""

    def __init__(self, m, n, k, nnz, *, s_rows=0.5, s_cols=0.5, seed=20260926, value_model=0, noise=0.1, rx=(1, 0, 1.0), ry=(1, 0, 1.0),
                 device=None, chunk=1 << 24):
        import torch
        self.torch = torch
        self.device = device if device is not None else torch.device("cpu")
        dev = self.device
        self.m, self.n, self.k = int(m), int(n), int(k)
        self.rows, self.cols = (0, self.m), (0, self.n)
        g = torch.Generator(device=dev)
        g.manual_seed(int(seed))
        f64 = dict(dtype=torch.float64, device=dev)
        # degrees and popularities
        wr = (torch.randperm(m, generator=g, device=dev).to(torch.float64) + 1.0) ** (-float(s_rows))
        wc = (torch.randperm(n, generator=g, device=dev).to(torch.float64) + 1.0) ** (-float(s_cols))
        cdf = torch.cumsum(wc / wc.sum(), 0)
        # (row, column) keys in chunks of rows: sample, sort, drop duplicates.  Heavy rows draw popular columns more than once and the
        # degree cap cuts the very heaviest, so the first pass falls short of `nnz`: the draw count is scaled up and the pass repeated
        # (at most twice) until the distinct entries are within 3 % of the target.
        draw = float(nnz)
        for attempt in range(3):
            deg = torch.clamp(torch.round(draw * wr / wr.sum()), 1, max(1, n // 4)).to(torch.int64)
            total = int(deg.sum().item())
            keys = []
            rptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
            rptr[1:] = torch.cumsum(deg, 0)
            r0 = 0
            while r0 < m:
                r1 = int(torch.searchsorted(rptr, torch.tensor([int(rptr[r0].item()) + chunk], device=dev), right=True).item()) - 1
                r1 = min(max(r1, r0 + 1), m)
                rid = torch.repeat_interleave(torch.arange(r0, r1, dtype=torch.int64, device=dev), deg[r0:r1])
                u = torch.rand(rid.numel(), generator=g, **f64)
                cid = torch.clamp(torch.searchsorted(cdf, u), max=n - 1)
                keys.append(torch.unique(rid * n + cid))  # sorted, duplicates dropped
                r0 = r1
            got = sum(int(t.numel()) for t in keys)
            if got >= 0.97 * nnz or attempt == 2:
                break
            draw *= min(3.0, nnz / max(got, 1)) * 1.02
            del keys
        key = torch.cat(keys) if len(keys) > 1 else keys[0]
        del keys
        rid, cid = key // n, key % n
        self.nnz_rows = self.nnz_cols = int(key.numel())
        self.sampled_with_duplicates = total
        self.rowptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        self.rowptr[1:] = torch.cumsum(torch.bincount(rid, minlength=m), 0)
        self.colidx = cid.to(torch.int32)
        # values: a = x*_e . y*_f + noise N(0,1), truth factors N(0, 1/sqrt(k)) (value_model 1: U(0,1)/sqrt(k), non-negative)
        if value_model == 1:
            Xs, Ys = torch.rand(m, k, generator=g, **f64) / k ** 0.5, torch.rand(n, k, generator=g, **f64) / k ** 0.5
        else:
            Xs, Ys = torch.randn(m, k, generator=g, **f64) / k ** 0.25, torch.randn(n, k, generator=g, **f64) / k ** 0.25
        vals = torch.empty(self.nnz_rows, **f64)
        for a in range(0, self.nnz_rows, chunk):
            b = min(a + chunk, self.nnz_rows)
            vals[a:b] = (Xs[rid[a:b]] * Ys[cid[a:b]]).sum(1) + noise * torch.randn(b - a, generator=g, **f64)
        del Xs, Ys
        self.rowvals = vals
        # the column view: the same entries ordered by (column, row)
        order = torch.argsort(cid * m + rid)
        self.rowidx = rid[order].to(torch.int32)
        self.colvals = vals[order]
        self.colptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        self.colptr[1:] = torch.cumsum(torch.bincount(cid, minlength=n), 0)
        del order, rid, cid, key
        self.max_row_len = int((self.rowptr[1:] - self.rowptr[:-1]).max().item())
        self.max_col_len = int((self.colptr[1:] - self.colptr[:-1]).max().item())
        self.losses = np.array([QUAD], dtype=_capi.LOSS_DTYPE)
        self.rx = np.array([rx], dtype=_capi.REG_DTYPE)
        self.ry = np.array([ry], dtype=_capi.REG_DTYPE)
        self._gen = g

    def whole_signature(self) -> "_capi.CSignature":
        sig = _capi.CSignature()
        sig.nnz_rows = sig.nnz_cols = self.nnz_rows
        sig.max_row_len, sig.max_col_len = self.max_row_len, self.max_col_len
        return sig

    def list_bytes(self) -> int:
        return (self.nnz_rows + self.nnz_cols) * 12

    def problem(self, borrow=False) -> _capi.ProblemArrays:
        if self.device.type != "cuda":
            return self.host_problem()
        p = lambda t: int(t.data_ptr())
        flags = _capi.PROBLEM_DEVICE_ARRAYS | (_capi.PROBLEM_BORROW_DEVICE_ARRAYS if borrow else 0)
        return _capi.ProblemArrays(self.m, self.n, self.k, p(self.rowptr), p(self.colidx), p(self.rowvals), p(self.colptr), p(self.rowidx),
                                   p(self.colvals), self.losses, self.rx, self.ry, 0, self.m, 0, self.n, flags=flags)

    def host_problem(self) -> _capi.ProblemArrays:
        f = lambda t: np.ascontiguousarray(t.cpu().numpy())
        return _capi.ProblemArrays(self.m, self.n, self.k, f(self.rowptr), f(self.colidx), f(self.rowvals), f(self.colptr), f(self.rowidx),
                                   f(self.colvals), self.losses, self.rx, self.ry)

    def init_factors(self, ld, init_seed=1):
        """X0 (ld x m) and Y0 (ld x n) iid N(0,1), padding rows zero (flat, like DeviceWorkload.init_factors)."""
        torch = self.torch
        g = torch.Generator(device=self.device)
        g.manual_seed(int(init_seed) + 7919)
        X = torch.zeros(self.m, ld, dtype=torch.float64, device=self.device)
        Y = torch.zeros(self.n, ld, dtype=torch.float64, device=self.device)
        X[:, : self.k] = torch.randn(self.m, self.k, generator=g, dtype=torch.float64, device=self.device)
        Y[:, : self.k] = torch.randn(self.n, self.k, generator=g, dtype=torch.float64, device=self.device)
        return X.reshape(-1), Y.reshape(-1)

    def degree_summary(self):
        torch = self.torch
        rl = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float64)
        cl = (self.colptr[1:] - self.colptr[:-1]).to(torch.float64)
        q = lambda t, p: float(torch.quantile(t[:: max(1, t.numel() // 1_000_000)], p).item())
        return {"observations": self.nnz_rows, "rows": {"mean": float(rl.mean().item()), "median": q(rl, 0.5), "p99": q(rl, 0.99), "max": self.max_row_len},
                "cols": {"mean": float(cl.mean().item()), "median": q(cl, 0.5), "p99": q(cl, 0.99), "max": self.max_col_len}}

    def free_sources(self):
        self.rowptr = self.colidx = self.rowvals = self.colptr = self.rowidx = self.colvals = None
        if self.device.type == "cuda":
            self.torch.cuda.empty_cache()
