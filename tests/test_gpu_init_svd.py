"""-m gpu: glrm_hip_init_svd (randomized subspace iteration on the resident lists) against the oracle's exact dense SVD."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi
from test_init_svd import model_of, numpy_init_svd

pytestmark = pytest.mark.gpu


def hip():
    return _capi.hip_api()


@pytest.mark.parametrize("name", ["scalar", "constant_and_empty_columns", "categorical_mix", "ordinal_offsets", "loss_test", "mnl"])
def test_init_svd_matches_exact_svd(name):
    g = model_of(name)
    Xn, Yn, Sn = numpy_init_svd(g)
    L.init_svd_(g, engine=hip(), tol=1e-13, max_iter=400)
    info = g._init_svd_info
    np.testing.assert_allclose(info["singular_values"], Sn, rtol=1e-8)
    P, Pn = g.X.T @ g.Y, Xn.T @ Yn
    assert np.abs(P - Pn).max() < 1e-5 * np.abs(Pn).max()           # north-star tolerance on the quantity that is unique
    go = model_of(name)
    L.init_svd_(go, engine=O.oracle_api())
    assert cases.fro_err(np.abs(g.X), np.abs(go.X)) < 1e-5 and cases.fro_err(np.abs(g.Y), np.abs(go.Y)) < 1e-5


def test_low_rank_model_converges_in_a_few_iterations():
    rng = np.random.default_rng(2)
    m, n, k = 3000, 400, 6
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) + 0.01 * rng.standard_normal((m, n))
    I, J = np.nonzero(rng.random((m, n)) < 0.5)
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), k, obs=(I, J))
    L.init_svd_(g, engine=hip(), tol=1e-9)
    Xn, Yn, Sn = numpy_init_svd(g)
    assert g._init_svd_info["iterations"] <= 30
    np.testing.assert_allclose(g._init_svd_info["singular_values"], Sn, rtol=1e-7)
    assert np.abs(g.X.T @ g.Y - Xn.T @ Yn).max() < 1e-5 * np.abs(Xn.T @ Yn).max()


def test_long_columns_and_wide_blocks():
    """Column lists longer than one 16384-entry chunk (partials in chunk order) and a block wider than 64 (two values per lane)."""
    rng = np.random.default_rng(4)
    m, n, k = 40000, 12, 3
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) + 0.05 * rng.standard_normal((m, n))
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), k)
    L.init_svd_(g, engine=hip(), tol=1e-10)
    Xn, Yn, Sn = numpy_init_svd(g)
    np.testing.assert_allclose(g._init_svd_info["singular_values"], Sn, rtol=1e-8)
    m, n, k = 300, 200, 70
    A = rng.standard_normal((m, k)) @ (rng.standard_normal((k, n)) * np.linspace(3, 1, k)[:, None]) + 0.01 * rng.standard_normal((m, n))
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), k)
    L.init_svd_(g, engine=hip(), tol=1e-11, max_iter=300)
    Xn, Yn, Sn = numpy_init_svd(g)
    np.testing.assert_allclose(g._init_svd_info["singular_values"], Sn, rtol=1e-6)


def test_rejects_dense_and_sharded_handles():
    rng = np.random.default_rng(6)
    g = L.GLRM(rng.standard_normal((64, 48)), L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), 16)
    api = hip()
    h = api.create(g.problem_arrays(dense=True))
    try:
        with pytest.raises(_capi.GLRMError):
            api.init_svd(h, np.zeros((16, 64), order="F"), np.zeros((16, 48), order="F"))
    finally:
        api.destroy(h)
    h = api.create(g.problem_arrays(rows=(0, 32)))
    try:
        with pytest.raises(_capi.GLRMError):
            api.init_svd(h, np.zeros((16, 64), order="F"), np.zeros((16, 48), order="F"))
    finally:
        api.destroy(h)
