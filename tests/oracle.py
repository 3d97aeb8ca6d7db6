"""Test-side access to the CPU oracle (oracle/libglrm_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module: the
oracle is the parity checker, never part of the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libglrm_oracle.so")
SANITIZED_LIB = os.environ.get("GLRM_ORACLE_LIB")  # `make -C oracle asan-check` runs the CPU suite on the ASan / UBSan build


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("glrm_oracle.c", "synth.c")] + [
        os.path.join(ROOT, "include", f) for f in ("glrm_hip.h", "glrm_synth.h")]
    stale = (not os.path.exists(ORACLE_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)
    return ORACLE_LIB


_api = None
_lib = None


def oracle_lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(SANITIZED_LIB if SANITIZED_LIB else build_oracle())
        _lib.glrm_cpu_loss_evaluate.restype = C.c_double
        _lib.glrm_cpu_loss_evaluate.argtypes = [C.c_void_p, C.c_double, C.c_double]
        _lib.glrm_cpu_loss_grad.restype = C.c_double
        _lib.glrm_cpu_loss_grad.argtypes = [C.c_void_p, C.c_double, C.c_double]
        _lib.glrm_cpu_reg_evaluate.restype = C.c_double
        _lib.glrm_cpu_reg_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib.glrm_cpu_reg_prox.restype = None
        _lib.glrm_cpu_reg_prox.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double]
        _lib.glrm_cpu_vloss_evaluate.restype = C.c_double
        _lib.glrm_cpu_vloss_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        _lib.glrm_cpu_vloss_grad.restype = None
        _lib.glrm_cpu_vloss_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        _lib.glrm_cpu_reg_evaluate_block.restype = C.c_double
        _lib.glrm_cpu_reg_evaluate_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib.glrm_cpu_reg_prox_block.restype = None
        _lib.glrm_cpu_reg_prox_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double]
        _lib.glrm_cpu_impute_entry.restype = C.c_double
        _lib.glrm_cpu_impute_entry.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        _lib.glrm_cpu_set_threads.argtypes = [C.c_int]
        _lib.glrm_cpu_set_dense_faithful.argtypes = [C.c_void_p, C.c_int]
        _lib.glrm_cpu_get_stepsizes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.glrm_cpu_set_sum_order.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        _lib.glrm_cpu_set_accept_bias.argtypes = [C.c_void_p, C.c_double]
    return _lib


def oracle_api():
    """The oracle behind the same Api binder the product uses for the HIP library."""
    global _api
    if _api is None:
        from lowrankmodels.jl_amd import _capi
        _api = _capi.Api(oracle_lib(), "glrm_cpu_", "cpu")
    return _api


def set_sum_order(h, which, order):
    """Make the oracle add the terms of the row (which = 0) / column (1) half-step in an ENGINE order: `order` is the CSumOrder
    glrm_hip_sum_order reported for a handle of the HIP engine (or one built by hand); None returns to the reference order."""
    api = oracle_api()
    api._ck(oracle_lib().glrm_cpu_set_sum_order(h, int(which), C.byref(order) if order is not None else None))


def set_dot_bias(bias):
    """Oracle-only test knob (process-wide): every dot product <x_e, y_f> times (1 + bias); 0 restores the reference's value."""
    lib = oracle_lib()
    lib.glrm_cpu_set_dot_bias.argtypes = [C.c_double]
    lib.glrm_cpu_set_dot_bias.restype = None
    lib.glrm_cpu_set_dot_bias(float(bias))


def set_accept_bias(h, bias):
    """Test knob: the line search accepts iff new < old + bias * |old| (0 = the reference's strict `<`).  Runs with +eps and -eps bracket
    every decision that hangs on the last bits of the two sums (oracle/glrm_oracle.c: accept_test)."""
    oracle_api()._ck(oracle_lib().glrm_cpu_set_accept_bias(h, float(bias)))


def make_sum_order(family, lanes, comps, waves=0, cached_maxlen=-1, cached_waves=0, batch=0, batch_one_wave_only=0, rotate=0, window=0,
                   windows_per_sup=0):
    """A glrm_sum_order by hand (CPU tests; on the GPU the engine reports its own through glrm_hip_sum_order)."""
    from lowrankmodels.jl_amd import _capi
    fam = {"reference": 0, "strided": 1, "windowed": 2}[family]
    return _capi.CSumOrder(fam, lanes, comps, waves, 1536, 98304, cached_maxlen, cached_waves, batch or (1 if fam == 1 else 2), batch_one_wave_only, rotate,
                           window, windows_per_sup, 0, 0)


def set_threads(n):
    oracle_lib().glrm_cpu_set_threads(int(n))


def usable_cores():
    """CPUs this process may actually run on: the smaller of the affinity mask and the cgroup CPU quota (a container
    that sees 256 hardware threads under a 16-CPU quota runs 256 OpenMP threads slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def _loss_struct(loss):
    from lowrankmodels.jl_amd import _capi
    k, r, s, p0, p1 = loss.descriptor()
    return _capi.CLoss(k, r, s, p0, p1)


def _reg_struct(reg):
    from lowrankmodels.jl_amd import _capi
    k, r, s = reg.descriptor()
    return _capi.CReg(k, r, s)


def loss_evaluate(loss, u, a):
    st = _loss_struct(loss)
    return oracle_lib().glrm_cpu_loss_evaluate(C.addressof(st), float(u), float(a))


def loss_grad(loss, u, a):
    st = _loss_struct(loss)
    return oracle_lib().glrm_cpu_loss_grad(C.addressof(st), float(u), float(a))


def vloss_evaluate(loss, u, a):
    """evaluate(l, u::Vector, a::Integer) of a multi-dimensional loss (a = level 1..max)."""
    st = _loss_struct(loss)
    u = np.array(u, dtype=np.float64)
    return oracle_lib().glrm_cpu_vloss_evaluate(C.addressof(st), u.ctypes.data, float(a))


def vloss_grad(loss, u, a):
    st = _loss_struct(loss)
    u = np.array(u, dtype=np.float64)
    g = np.zeros_like(u)
    oracle_lib().glrm_cpu_vloss_grad(C.addressof(st), u.ctypes.data, float(a), g.ctypes.data)
    return g


def impute_entry(domain, loss, u):
    """impute(D, l, u) of the oracle for one entry; raises TypeError for pairs the reference has no rule for."""
    from lowrankmodels.jl_amd import _capi
    st = _loss_struct(loss)
    dom = np.array([domain.descriptor()], dtype=_capi.DOMAIN_DTYPE)
    u = np.atleast_1d(np.asarray(u, dtype=np.float64)).copy()
    bad = C.c_int(0)
    v = oracle_lib().glrm_cpu_impute_entry(dom.ctypes.data, C.addressof(st), u.ctypes.data, C.byref(bad))
    if bad.value:
        raise TypeError("no impute rule")
    return v


def reg_evaluate_block(reg, a):
    """evaluate(r, a) for a k x d block (numpy (k, d)); wrappers included."""
    st = _reg_struct(reg)
    a = np.asfortranarray(np.atleast_2d(np.asarray(a, dtype=np.float64).T).T if np.ndim(a) == 1 else np.asarray(a, dtype=np.float64))
    k, d = (a.shape[0], 1) if a.ndim == 1 else a.shape
    return oracle_lib().glrm_cpu_reg_evaluate_block(C.addressof(st), a.ctypes.data, k, d)


def reg_prox_block(reg, u, alpha):
    st = _reg_struct(reg)
    u = np.array(u, dtype=np.float64, order="F")
    k, d = (u.shape[0], 1) if u.ndim == 1 else u.shape
    oracle_lib().glrm_cpu_reg_prox_block(C.addressof(st), u.ctypes.data, k, d, float(alpha))
    return u


def reg_evaluate(reg, x):
    st = _reg_struct(reg)
    x = np.ascontiguousarray(x, dtype=np.float64)
    return oracle_lib().glrm_cpu_reg_evaluate(C.addressof(st), x.ctypes.data, x.size)


def reg_prox(reg, u, alpha):
    st = _reg_struct(reg)
    u = np.array(u, dtype=np.float64)
    oracle_lib().glrm_cpu_reg_prox(C.addressof(st), u.ctypes.data, u.size, float(alpha))
    return u


# ------------------------------------------------------------------ synthetic workloads (CPU)

class SynthSpec(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("k", C.c_int32), ("q", C.c_int32), ("seed", C.c_uint64),
                ("value_model", C.c_int32), ("loss_mix", C.c_int32), ("noise", C.c_double)]


def synth_cpu_init(m, n, k, q, seed=20260926, init_seed=1):
    """Only the start X0 (k x m), Y0 (k x n) of synth_cpu (the generator's Box-Muller normals through THIS host's libm: the device
    generator's differ from them in the last bits, so a bit-for-bit comparison with an oracle run hands these to the engine)."""
    lib = oracle_lib()
    s = SynthSpec(m, n, k, q, seed, 0, 0, 0.1)
    X0 = np.zeros((k, m), order="F")
    Y0 = np.zeros((k, n), order="F")
    assert lib.glrm_synth_cpu_init(C.byref(s), C.c_uint64(init_seed), C.c_int(k), C.c_void_p(X0.ctypes.data), C.c_void_p(Y0.ctypes.data)) == 0
    return X0, Y0


def synth_cpu(m, n, k, q, seed=20260926, value_model=0, loss_mix=0, noise=0.1, init_seed=1,
              rows=None, cols=None, transpose=False):
    """Generate (rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0) with oracle/synth.c.
    transpose=True (whole problems only): the column view is counting-sorted out of the row view (O(nnz)) instead of the generator's
    O(n x m) hash scan -- the same arrays (tests/test_synth.py), what bench.py's CPU legs use."""
    lib = oracle_lib()
    s = SynthSpec(m, n, k, q, seed, value_model, loss_mix, noise)
    rb, re = (0, m) if rows is None else rows
    cb, ce = (0, n) if cols is None else cols
    nzr = (re - rb) * q
    rowptr = np.zeros(re - rb + 1, np.int64)
    colidx = np.zeros(nzr, np.int32)
    rowvals = np.zeros(nzr, np.float64)
    assert lib.glrm_synth_cpu_rows(C.byref(s), C.c_int64(rb), C.c_int64(re), C.c_void_p(rowptr.ctypes.data),
                                   C.c_void_p(colidx.ctypes.data), C.c_void_p(rowvals.ctypes.data)) == 0
    if transpose:
        assert (rb, re, cb, ce) == (0, m, 0, n), "transpose=True builds whole problems"
        colptr = np.zeros(n + 1, np.int64)
        rowidx = np.zeros(nzr, np.int32)
        colvals = np.zeros(nzr, np.float64)
        assert lib.glrm_synth_cpu_cols_from_rows(C.c_int64(m), C.c_int64(0), C.c_int64(n), C.c_void_p(rowptr.ctypes.data), C.c_void_p(colidx.ctypes.data),
                                                 C.c_void_p(rowvals.ctypes.data), C.c_void_p(colptr.ctypes.data), C.c_void_p(rowidx.ctypes.data),
                                                 C.c_void_p(colvals.ctypes.data)) == 0
        X0 = np.zeros((k, m), order="F")
        Y0 = np.zeros((k, n), order="F")
        assert lib.glrm_synth_cpu_init(C.byref(s), C.c_uint64(init_seed), C.c_int(k), C.c_void_p(X0.ctypes.data), C.c_void_p(Y0.ctypes.data)) == 0
        return rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0
    colptr = np.zeros(ce - cb + 1, np.int64)
    assert lib.glrm_synth_cpu_col_counts(C.byref(s), C.c_int64(cb), C.c_int64(ce), C.c_void_p(colptr.ctypes.data)) == 0
    nzc = int(colptr[-1])
    rowidx = np.zeros(nzc, np.int32)
    colvals = np.zeros(nzc, np.float64)
    assert lib.glrm_synth_cpu_cols(C.byref(s), C.c_int64(cb), C.c_int64(ce), C.c_void_p(colptr.ctypes.data),
                                   C.c_void_p(rowidx.ctypes.data), C.c_void_p(colvals.ctypes.data)) == 0
    X0 = np.zeros((k, m), order="F")
    Y0 = np.zeros((k, n), order="F")
    assert lib.glrm_synth_cpu_init(C.byref(s), C.c_uint64(init_seed), C.c_int(k), C.c_void_p(X0.ctypes.data),
                                   C.c_void_p(Y0.ctypes.data)) == 0
    return rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0
