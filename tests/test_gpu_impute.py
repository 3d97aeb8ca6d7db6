"""-m gpu: glrm_hip_impute / glrm_hip_error_metric against the oracle (same domains, same factors)."""
import numpy as np
import pytest

import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi
from test_impute import heterogeneous_model

pytestmark = pytest.mark.gpu


def hip():
    return _capi.hip_api()


def test_impute_and_error_metric_match_oracle():
    rng = np.random.default_rng(4)
    losses, X, Y, k = heterogeneous_model(rng, 500)
    m = X.shape[1]
    g0 = L.GLRM(np.ones((m, len(losses))), losses, L.ZeroReg(), L.ZeroReg(), k, X=X, Y=Y)
    A_c = L.impute(g0, engine=O.oracle_api())
    g0.close()
    A_g = L.impute(g0, engine=hip())
    # dots differ in the last bits only; rounding to levels can flip at exact ties, which random data does not produce
    assert np.array_equal(A_c[:, 4:10], A_g[:, 4:10])                          # ordinal / boolean / categorical columns: exact
    np.testing.assert_allclose(A_g, A_c, rtol=1e-12, atol=1e-13)
    I, J = np.nonzero(rng.random(A_c.shape) < 0.7)
    g = L.GLRM(A_c, losses, L.ZeroReg(), L.ZeroReg(), k, obs=(I, J), X=X, Y=Y)
    assert L.error_metric(g, engine=hip()) == 0.0 and L.error_metric(g, standardize=True, engine=hip()) == 0.0   # test/err_test.jl:49
    Xp = X + 0.3 * rng.standard_normal(X.shape)
    for std in (False, True):
        g.close()
        e_c = L.error_metric(g, Xp, Y, standardize=std, engine=O.oracle_api())
        g.close()
        e_g = L.error_metric(g, Xp, Y, standardize=std, engine=hip())
        assert e_g == pytest.approx(e_c, rel=1e-10) and e_c > 0
    doms = [L.default_domain(l) for l in losses]
    doms[0] = L.OrdinalDomain(-3, 3); doms[1] = L.BoolDomain(); doms[7] = L.OrdinalDomain(1, 4)   # non-default pairings
    g.close()
    e_c = L.error_metric(g, Xp, Y, doms, engine=O.oracle_api())
    g.close()
    assert L.error_metric(g, Xp, Y, doms, engine=hip()) == pytest.approx(e_c, rel=1e-10)


def test_long_columns_and_unsupported_pairs():
    rng = np.random.default_rng(8)
    m, n, k = 150000, 3, 5
    A = np.round(rng.standard_normal((m, k)) @ rng.standard_normal((k, n)))
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(), L.QuadReg(), k, rng=rng)
    doms = [L.OrdinalDomain(-4, 4)] * n
    e_g = L.error_metric(g, domains=doms, standardize=True, engine=hip())     # columns longer than one 65536-entry chunk
    g.close()
    assert e_g == pytest.approx(L.error_metric(g, domains=doms, standardize=True, engine=O.oracle_api()), rel=1e-10)
    gl = L.GLRM(A[:100] > 0, L.LogisticLoss(), L.QuadReg(), L.QuadReg(), 2, rng=rng)
    with pytest.raises(L.GLRMError) as ei:
        L.impute(gl, domains=[L.RealDomain()] * n, engine=hip())
    assert ei.value.code == _capi.ERR_UNSUPPORTED
