"""glrm_*_multi_* (one process, N shards; include/glrm_hip.h): the library-side sharding, exchange bookkeeping and outer loop give
the single-shard bits for every shard count.  CPU: the oracle's twin (shared buffers).  -m gpu: the HIP engine with several shards
on ONE device (device_ids repeat; the peer pushes become device-to-device copies) against the single-device fit and the oracle."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi

GOLDEN_NAMES = ["c1", "c4", "nnmf", "mixed", "kmeans", "mnl_ordinal", "loss_test"]


def run_multi(api, pa, X0, Y0, params, n_shards, **kw):
    mh = api.multi_create(pa, n_shards, **kw)
    try:
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        obj, sec = api.multi_fit(mh, params, X, Y)
        info = api.multi_info(mh, n_shards)
    finally:
        api.multi_destroy(mh)
    return obj, X, Y, info


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_oracle_multi_equals_single(name):
    kwargs, params = cases.build_golden_case(name)
    pa = L.GLRM(**kwargs).problem_arrays()
    O.set_threads(1)
    api = O.oracle_api()
    o1, X1, Y1, _ = cases.run_engine(api, pa, kwargs["X"], kwargs["Y"], params)
    for n in (1, 2, 3, 8):
        o, X, Y, info = run_multi(api, pa, kwargs["X"], kwargs["Y"], params, n)
        assert len(o) == len(o1) and np.array_equal(o[1:], o1[1:]), (name, n)
        assert cases.rel_err(o[:1], o1[:1]) < 1e-12       # initial objective: per-column sums first
        assert np.array_equal(X, X1) and np.array_equal(Y, Y1), (name, n)
        assert info["row_bounds"][0] == 0 and info["row_bounds"][-1] == pa.m and info["col_bounds"][-1] == pa.n
        assert info["row_bounds"] == L.partition(pa.rowptr, n) and info["col_bounds"] == L.partition(pa.colptr, n)


def test_oracle_multi_host_level_and_regularizer_swap():
    """HipProxGradParams(ngpus=N) through fit!: cached multi handle, warm start, scale_regularizer! keeps the handle."""
    kwargs, params = cases.build_golden_case("c1")
    api = O.oracle_api()
    g1, g2 = L.GLRM(**kwargs), L.GLRM(**{**kwargs, "X": kwargs["X"].copy(), "Y": kwargs["Y"].copy()})
    p1, p2 = L.HipProxGradParams(max_iter=6), L.HipProxGradParams(max_iter=6, ngpus=3)
    for g, p in ((g1, p1), (g2, p2)):
        L.fit_b(g, p, verbose=False, engine=api)
        L.scale_regularizer_(g, 0.5)
        L.fit_b(g, p, verbose=False, engine=api)
    assert len(g2._handle_cache) == 6 and g2._handle_cache[5] == "multi"
    assert np.array_equal(g1.X, g2.X) and np.array_equal(g1.Y, g2.Y)
    g1.close(); g2.close()


def test_multi_argument_checks():
    kwargs, params = cases.build_golden_case("c1")
    pa = L.GLRM(**kwargs).problem_arrays()
    api = O.oracle_api()
    with pytest.raises(_capi.GLRMError):
        api.multi_create(pa, 0)
    shard = _capi.ProblemArrays(pa.m, pa.n, pa.k, pa.rowptr, pa.colidx, pa.rowvals, pa.colptr, pa.rowidx, pa.colvals, pa.losses, pa.rx,
                                pa.ry, row_begin=0, row_end=50)
    with pytest.raises(_capi.GLRMError):
        api.multi_create(shard, 2)
    with pytest.raises(ValueError):
        L.HipProxGradParams(ngpus=2, device_ids=[0])


# ------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_hip_multi_shards_on_one_device_equal_single(name):
    kwargs, params = cases.build_golden_case(name)
    pa = L.GLRM(**kwargs).problem_arrays()
    api = _capi.hip_api()
    o1, X1, Y1, _ = cases.run_engine(api, pa, kwargs["X"], kwargs["Y"], params)
    for n, chunks in ((2, 0), (3, 2), (8, 0)):
        o, X, Y, info = run_multi(api, pa, kwargs["X"], kwargs["Y"], params, n, device_ids=[0] * n, x_chunks=chunks)
        assert len(o) == len(o1) and np.array_equal(o[1:], o1[1:]), (name, n)
        assert cases.rel_err(o[:1], o1[:1]) < 1e-12
        assert np.array_equal(X, X1) and np.array_equal(Y, Y1), (name, n)
        assert info["exchange"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("tiled", [1, 2])
def test_hip_multi_c4_recipe_vs_oracle(tiled):
    """The north-star recipe on 4 shards of one device (gather and LDS-tiled sweeps, pipelined X exchange) against the oracle."""
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(20000, 2000, 64, 100, value_model=1)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(3, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(20000, 2000, 64, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    X0, Y0 = np.asfortranarray(np.abs(X0) / 8.0), np.asfortranarray(np.abs(Y0) / 8.0)
    params = L.ProxGradParams(max_iter=8)
    O.set_threads(4)
    o_c, X_c, Y_c, _ = cases.run_engine(O.oracle_api(), pa, X0, Y0, params)
    o, X, Y, info = run_multi(_capi.hip_api(), pa, X0, Y0, params, 4, device_ids=[0] * 4, x_chunks=4, tiled=tiled, profile=1)
    assert len(o) == len(o_c) and cases.rel_err(o, o_c) < 1e-5
    assert cases.fro_err(X, X_c) < 1e-5 and cases.fro_err(Y, Y_c) < 1e-5
    assert info["exchange_ms"] >= 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("quad_gram", [False, True])
def test_hip_multi_dense_handover_and_host_level(quad_gram):
    """Fully observed QuadLoss: the dense hand-over (matrix cores) on two shards, through HipProxGradParams(ngpus=2); with
    glrm_options.quad_gram the Gram matrix of the replicated opposing factor is the same on every shard."""
    rng = np.random.default_rng(5)
    m, n, k = 300, 200, 16
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) / 4 + 0.1 * rng.standard_normal((m, n))
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    g1 = L.GLRM(A, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), k, X=X0.copy(), Y=Y0.copy())
    g2 = L.GLRM(A, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), k, X=X0.copy(), Y=Y0.copy())
    _, _, ch1 = L.fit_b(g1, L.HipProxGradParams(max_iter=10, quad_gram=quad_gram), verbose=False)
    _, _, ch2 = L.fit_b(g2, L.HipProxGradParams(max_iter=10, ngpus=2, device_ids=[0, 0], quad_gram=quad_gram), verbose=False)
    assert cases.rel_err(ch2.objective, ch1.objective) < 1e-9 and cases.fro_err(g2.X, g1.X) < 1e-9
    g1.close(); g2.close()


@pytest.mark.gpu
def test_hip_multi_rccl_single_device_falls_back_or_runs():
    """exchange=1 (RCCL) with one shard is a no-op exchange; with repeated devices the library must fall back to the direct path."""
    kwargs, params = cases.build_golden_case("c1")
    pa = L.GLRM(**kwargs).problem_arrays()
    api = _capi.hip_api()
    o1, X1, Y1, _ = cases.run_engine(api, pa, kwargs["X"], kwargs["Y"], params)
    for n in (1, 2):
        o, X, Y, info = run_multi(api, pa, kwargs["X"], kwargs["Y"], params, n, device_ids=[0] * n, exchange=1)
        assert np.array_equal(o[1:], o1[1:]) and np.array_equal(X, X1)
        assert info["exchange"] == 0
