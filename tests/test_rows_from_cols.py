"""GLRM_PROBLEM_ROWS_FROM_COLS (include/glrm_hip.h): Omega as a sparse matrix's pattern handed over as its column view only
(colptr / rowval / nzval of a CSC matrix after an index shift) -- the row view, every row's entries by ascending column like
`sort_observations` pushes the CartesianIndices of `findall(!iszero, A)` (src/glrm.jl:46-48, src/modify_glrm.jl:8-12), is derived by the
library.  An Omega beyond what one hipCUB sort takes (1.5e9 entries; C5 holds 5e9) is sorted in row ranges: GLRM_HIP_TRANSPOSE_CHUNK lowers that
bound so the ranged path runs here on small patterns.  CPU: the oracle's twin.  -m gpu: the engine's device-side derivation (one stable radix sort of the column-major stream by row
id) gives exactly the handle built from both views -- same factors bit for bit -- on sorted patterns, patterns with duplicates, empty
rows and columns, through the in-library multi-GPU create (host-side transpose there), and refuses what the flag does not cover."""
import numpy as np
import pytest
import scipy.sparse as sp

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi


def pattern(rng, m, n, k, density, dup=False):
    """A sparse matrix as CSC arrays (+ the row view a host would have to build) and a start."""
    A = sp.random(m, n, density=density, format="csc", random_state=np.random.RandomState(int(rng.integers(1 << 30))), data_rvs=lambda s: rng.standard_normal(s))
    A.data[A.data == 0] = 1.0
    colptr, rowidx, colvals = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float64)
    if dup:  # repeat some entries inside their column (duplicates are kept by both views, src/modify_glrm.jl:8-12)
        reps = 1 + (rng.random(len(rowidx)) < 0.1).astype(np.int64)
        col_of = np.repeat(np.arange(n), np.diff(colptr))
        rowidx, colvals, col_of = np.repeat(rowidx, reps), np.repeat(colvals, reps), np.repeat(col_of, reps)
        colptr = np.concatenate([[0], np.cumsum(np.bincount(col_of, minlength=n))]).astype(np.int64)
    # the host's counting transpose (what the flag saves): columns in order -> every row's list ascending
    order = np.argsort(rowidx, kind="stable")
    cols_of = np.repeat(np.arange(n, dtype=np.int32), np.diff(colptr))
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rowidx, minlength=m))]).astype(np.int64)
    colidx, rowvals = cols_of[order], colvals[order]
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 0.3)], dtype=_capi.REG_DTYPE)
    both = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    cols_only = _capi.ProblemArrays(m, n, k, None, None, None, colptr, rowidx, colvals, one, reg, reg, flags=_capi.PROBLEM_ROWS_FROM_COLS)
    X0, Y0 = np.asfortranarray(rng.standard_normal((k, m))), np.asfortranarray(rng.standard_normal((k, n)))
    return both, cols_only, X0, Y0


def same_fit(api, both, cols_only, X0, Y0, **kw):
    p = L.ProxGradParams(max_iter=8)
    a = cases.run_engine(api, both, X0, Y0, p, **kw)
    b = cases.run_engine(api, cols_only, X0, Y0, p, **kw)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[3]["nnz_rows"] == b[3]["nnz_rows"] == a[3]["nnz_cols"] and a[3]["tiled"] == b[3]["tiled"]
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert a[3][key] == b[3][key]


@pytest.mark.parametrize("dup", [False, True])
def test_oracle_twin(dup):
    rng = np.random.default_rng(3 + dup)
    same_fit(O.oracle_api(), *pattern(rng, 300, 70, 5, 0.2, dup=dup))


def test_oracle_twin_on_an_empty_pattern_and_a_single_entry():
    api = O.oracle_api()
    same_fit(api, *pattern(np.random.default_rng(1), 30, 12, 3, 0.0))       # no observation at all: the line searches run on the regularizers alone
    both, cols_only, X0, Y0 = pattern(np.random.default_rng(2), 1, 1, 2, 1.0)
    same_fit(api, both, cols_only, X0, Y0)


def test_oracle_twin_refuses_misuse():
    rng = np.random.default_rng(5)
    both, cols_only, X0, Y0 = pattern(rng, 60, 20, 3, 0.3)
    api = O.oracle_api()
    bad = _capi.ProblemArrays(60, 20, 3, both.rowptr, both.colidx, both.rowvals, both.colptr, both.rowidx, both.colvals, both.losses, both.rx, both.ry,
                              flags=_capi.PROBLEM_ROWS_FROM_COLS)
    with pytest.raises(_capi.GLRMError):      # the row arrays must be NULL with the flag
        api.create(bad)
    shard = _capi.ProblemArrays(60, 20, 3, None, None, None, both.colptr[:11], both.rowidx, both.colvals, both.losses, both.rx, both.ry,
                                col_begin=0, col_end=10, flags=_capi.PROBLEM_ROWS_FROM_COLS)
    with pytest.raises(_capi.GLRMError):      # whole problems only
        api.create(shard)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(300, 70, 5, 0.2, False), (5000, 400, 32, 0.05, False), (2000, 300, 16, 0.1, True), (40, 900, 8, 0.02, False)])
def test_device_side_row_view_gives_the_handle_built_from_both_views(shape):
    m, n, k, dens, dup = shape
    rng = np.random.default_rng(m + n)
    both, cols_only, X0, Y0 = pattern(rng, m, n, k, dens, dup=dup)
    api = _capi.hip_api()
    same_fit(api, both, cols_only, X0, Y0)
    same_fit(api, both, cols_only, X0, Y0, tiled=2)      # (LDS-tiled where the lists allow; the gather sweeps otherwise)
    o_c = cases.run_engine(O.oracle_api(), cols_only, X0, Y0, L.ProxGradParams(max_iter=8))
    o_g = cases.run_engine(api, cols_only, X0, Y0, L.ProxGradParams(max_iter=8))
    assert cases.rel_err(o_g[0], o_c[0]) < 1e-5 and cases.fro_err(o_g[1], o_c[1]) < 1e-5 and cases.fro_err(o_g[2], o_c[2]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [1, 97, 5000, 60000])
@pytest.mark.parametrize("shape", [(3000, 250, 8, 0.08, False), (700, 300, 5, 0.15, True), (40, 900, 8, 0.02, False)])
def test_row_ranges_give_the_same_row_view(shape, chunk, monkeypatch):
    """Row ranges of at most `chunk` entries (one row at least: chunk = 1 sorts row by row, 60000 = one or two ranges): the same handle."""
    m, n, k, dens, dup = shape
    rng = np.random.default_rng(m + n + 1)
    both, cols_only, X0, Y0 = pattern(rng, m, n, k, dens, dup=dup)
    if chunk == 1 and m > 1000:
        pytest.skip("row-by-row on the small patterns only")
    api = _capi.hip_api()
    p = L.ProxGradParams(max_iter=5)
    a = cases.run_engine(api, both, X0, Y0, p)
    monkeypatch.setenv("GLRM_HIP_TRANSPOSE_CHUNK", str(chunk))
    b = cases.run_engine(api, cols_only, X0, Y0, p)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[3]["nnz_rows"] == b[3]["nnz_rows"]


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [None, 7])
def test_device_side_row_view_of_an_empty_pattern_and_a_single_entry(chunk, monkeypatch):
    if chunk:
        monkeypatch.setenv("GLRM_HIP_TRANSPOSE_CHUNK", str(chunk))
    api = _capi.hip_api()
    same_fit(api, *pattern(np.random.default_rng(1), 30, 12, 3, 0.0))
    same_fit(api, *pattern(np.random.default_rng(2), 1, 1, 2, 1.0))
    same_fit(api, *pattern(np.random.default_rng(3), 1, 40, 2, 0.5))         # one row holding everything: a range of one row
    same_fit(api, *pattern(np.random.default_rng(4), 40, 1, 2, 0.5))


@pytest.mark.gpu
def test_device_arrays_multi_create_and_refusals():
    import torch
    from test_multi_in_process import run_multi
    rng = np.random.default_rng(9)
    both, cols_only, X0, Y0 = pattern(rng, 1500, 200, 8, 0.1)
    api = _capi.hip_api()
    p = L.ProxGradParams(max_iter=6)
    ref = cases.run_engine(api, both, X0, Y0, p)
    # the column view already in HBM (GLRM_PROBLEM_DEVICE_ARRAYS)
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(a).to(dev) for a in (cols_only.colptr, cols_only.rowidx, cols_only.colvals)]
    ondev = _capi.ProblemArrays(1500, 200, 8, None, None, None, *(int(x.data_ptr()) for x in t), both.losses, both.rx, both.ry,
                                flags=_capi.PROBLEM_ROWS_FROM_COLS | _capi.PROBLEM_DEVICE_ARRAYS)
    got = cases.run_engine(api, ondev, X0, Y0, p)
    assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    # the in-library multi-GPU create builds the row view on the host and shards both views
    o, X, Y, info = run_multi(api, cols_only, X0, Y0, p, 3, device_ids=[0, 0, 0])
    assert np.array_equal(o[1:], ref[0][1:]) and np.array_equal(X, ref[1]) and np.array_equal(Y, ref[2])
    # refusals
    with pytest.raises(_capi.GLRMError):
        api.create(_capi.ProblemArrays(1500, 200, 8, both.rowptr, both.colidx, both.rowvals, both.colptr, both.rowidx, both.colvals, both.losses, both.rx, both.ry,
                                       flags=_capi.PROBLEM_ROWS_FROM_COLS))
    with pytest.raises(_capi.GLRMError):
        api.create(_capi.ProblemArrays(1500, 200, 8, None, None, None, *(int(x.data_ptr()) for x in t), both.losses, both.rx, both.ry,
                                       flags=_capi.PROBLEM_ROWS_FROM_COLS | _capi.PROBLEM_DEVICE_ARRAYS | _capi.PROBLEM_BORROW_DEVICE_ARRAYS))
    badidx = cols_only.rowidx.copy()
    badidx[7] = 1500
    with pytest.raises(_capi.GLRMError):
        api.create(_capi.ProblemArrays(1500, 200, 8, None, None, None, cols_only.colptr, badidx, cols_only.colvals, both.losses, both.rx, both.ry,
                                       flags=_capi.PROBLEM_ROWS_FROM_COLS))


def test_multi_create_checks_the_column_view_before_it_transposes():
    """the host counting transpose of glrm_hip_multi_create ran BEFORE any structural check of colptr -- a colptr that does
    not start at 0, is not monotone, or ends beyond the arrays read and wrote past the vectors; a negative colptr[n] threw across the extern "C"
    boundary.  Every such view is now refused with GLRM_ERR_INVALID before anything is counted (no device is touched: this runs without a GPU,
    through the product library's own entry point)."""
    rng = np.random.default_rng(12)
    both, cols_only, X0, Y0 = pattern(rng, 200, 40, 4, 0.2)
    api = _capi.hip_api()

    def refused(colptr, rowidx=None):
        bad = _capi.ProblemArrays(200, 40, 4, None, None, None, np.ascontiguousarray(colptr, dtype=np.int64), cols_only.rowidx if rowidx is None else rowidx,
                                  cols_only.colvals, both.losses, both.rx, both.ry, flags=_capi.PROBLEM_ROWS_FROM_COLS)
        with pytest.raises(_capi.GLRMError) as e:
            api.multi_create(bad, 2, device_ids=[0, 0])
        assert e.value.code == _capi.ERR_INVALID, e.value

    cp = cols_only.colptr
    refused(cp + 1)                                                # does not start at 0 (and ends one past the arrays)
    swapped = cp.copy(); swapped[10], swapped[11] = cp[11] + 5, cp[10]
    refused(swapped)                                               # not monotone
    neg = cp.copy(); neg[-1] = -1
    refused(neg)                                                   # negative total: used to throw std::length_error through the C ABI
    badidx = cols_only.rowidx.copy(); badidx[3] = 200
    refused(cp, badidx)                                            # row index outside [0, m)
