"""CPU generator of the synthetic workloads (include/glrm_synth.h): Omega is stratified (sorted, unique, q per
row), the CSR and CSC views describe the same set with the same values, shards concatenate to the whole."""
import numpy as np

import oracle as O


def test_stratified_omega_and_views_agree():
    m, n, k, q = 300, 120, 4, 12
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, seed=5, loss_mix=1)
    assert np.array_equal(rowptr, np.arange(m + 1) * q)
    S = n // q
    ci = colidx.reshape(m, q)
    assert np.all(np.diff(ci, axis=1) > 0) and np.all(ci // S == np.arange(q))  # one per stratum, sorted, unique
    assert colptr[-1] == m * q
    rows_of_csr = np.repeat(np.arange(m), q)
    a = set(zip(rows_of_csr.tolist(), colidx.tolist()))
    cols_of_csc = np.repeat(np.arange(n), np.diff(colptr))
    b = set(zip(rowidx.tolist(), cols_of_csc.tolist()))
    assert a == b and len(a) == m * q
    for f in range(n):
        assert np.all(np.diff(rowidx[colptr[f]:colptr[f + 1]]) > 0)  # ascending rows (findall order)
    va = dict(zip(zip(rows_of_csr.tolist(), colidx.tolist()), rowvals.tolist()))
    vb = dict(zip(zip(rowidx.tolist(), cols_of_csc.tolist()), colvals.tolist()))
    assert va == vb
    # loss_mix: f%3==1 -> labels in {0,1}; f%3==2 -> ordinal levels 1..5
    kinds = cols_of_csc % 3
    assert set(np.unique(colvals[kinds == 1]).tolist()) <= {0.0, 1.0}
    assert set(np.unique(colvals[kinds == 2]).tolist()) <= {1.0, 2.0, 3.0, 4.0, 5.0}
    assert abs(X0.std() - 1) < 0.1 and abs(Y0.std() - 1) < 0.15


def test_shards_concatenate_to_the_whole():
    m, n, k, q = 200, 60, 3, 6
    full = O.synth_cpu(m, n, k, q, seed=9)
    a = O.synth_cpu(m, n, k, q, seed=9, rows=(0, 80), cols=(0, 25))
    b = O.synth_cpu(m, n, k, q, seed=9, rows=(80, 200), cols=(25, 60))
    assert np.array_equal(np.concatenate([a[1], b[1]]), full[1]) and np.array_equal(np.concatenate([a[2], b[2]]), full[2])
    assert np.array_equal(np.concatenate([a[4], b[4]]), full[4]) and np.array_equal(np.concatenate([a[5], b[5]]), full[5])
    assert np.array_equal(np.concatenate([a[3], a[3][-1] + b[3][1:]]), full[3])


def test_column_view_by_transposition_equals_the_generated_one():
    """bench.py's CPU legs build the column view by a stable counting sort of the row view (O(nnz)); it is the generator's own."""
    for mix in (0, 1):
        a = O.synth_cpu(700, 90, 5, 9, seed=3, loss_mix=mix)
        b = O.synth_cpu(700, 90, 5, 9, seed=3, loss_mix=mix, transpose=True)
        for u, v in zip(a, b):
            assert np.array_equal(u, v)


def test_zipf_workload_is_a_consistent_power_law_omega():
    """lowrankmodels.jl_amd/synth.py: ZipfWorkload (pure torch, so it runs here): both views list the same entries, sorted and without
    duplicates (the findall order of src/glrm.jl:46-48), row degrees and column popularities are heavy-tailed, the values carry a rank-k
    signal."""
    import scipy.sparse as sp
    from lowrankmodels.jl_amd import synth
    m, n, k = 6000, 800, 8
    w = synth.ZipfWorkload(m, n, k, 400_000, s_rows=0.7, s_cols=0.7, seed=3, chunk=1 << 17)
    pa = w.host_problem()
    assert pa.rowptr[-1] == pa.colptr[-1] == w.nnz_rows and 0.85 * 400_000 < w.nnz_rows < 1.1 * 400_000
    for ptr, idx in ((pa.rowptr, pa.colidx), (pa.colptr, pa.rowidx)):
        d = np.diff(idx.astype(np.int64))
        inner = np.ones(len(idx) - 1, dtype=bool)
        inner[ptr[1:-1][(ptr[1:-1] > 0) & (ptr[1:-1] < len(idx))] - 1] = False   # differences across a segment boundary
        assert (d[inner] > 0).all()
    A = sp.csr_matrix((pa.rowvals, pa.colidx, pa.rowptr), shape=(m, n))
    B = sp.csc_matrix((pa.colvals, pa.rowidx, pa.colptr), shape=(m, n))
    assert abs(A - B.tocsr()).max() == 0.0
    s = w.degree_summary()
    assert s["rows"]["max"] > 2.5 * s["rows"]["median"] and s["cols"]["max"] > 10 * s["cols"]["median"], s   # (800 columns cap the heaviest rows)
    sig = w.whole_signature()
    assert sig.nnz_rows == w.nnz_rows and sig.max_col_len == s["cols"]["max"]
    X0, Y0 = w.init_factors(8)
    assert X0.numel() == m * 8 and Y0.numel() == n * 8
