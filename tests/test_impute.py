"""impute / error_metric (src/impute_and_err.jl, src/evaluate_fit.jl:107-168): oracle vs the Python mirrors entry by entry, the
reference's own consistency property (test/err_test.jl:33-51: data imputed from a low-rank model has error_metric == 0 at that
model), and the model-level functions on the oracle engine."""
import math

import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O

SCALAR = [L.QuadLoss(), L.L1Loss(2.0), L.HuberLoss(), L.QuantileLoss(quantile=0.3), L.PeriodicLoss(2.5), L.PoissonLoss(20),
          L.OrdinalHingeLoss(1, 10), L.LogisticLoss(), L.WeightedHingeLoss(1.0, case_weight_ratio=2.0)]
VECTOR = [L.MultinomialLoss(4), L.OvALoss(3), L.OvALoss(4, bin_loss=L.HingeLoss()), L.BvSLoss(5), L.OrdisticLoss(4), L.MultinomialOrdinalLoss(5)]
DOMAINS = [L.RealDomain(), L.BoolDomain(), L.OrdinalDomain(1, 7), L.PeriodicDomain(2.5), L.CountDomain(12), L.CategoricalDomain(4)]


def test_known_answers():
    assert L.impute_entry(L.RealDomain(), L.QuadLoss(), 1.7) == 1.7
    assert L.impute_entry(L.OrdinalDomain(1, 5), L.QuadLoss(), 7.2) == 5 and L.impute_entry(L.OrdinalDomain(1, 5), L.QuadLoss(), 2.5) == 2  # half to even
    assert L.impute_entry(L.BoolDomain(), L.LogisticLoss(), 0.0) is True and L.impute_entry(L.BoolDomain(), L.LogisticLoss(), -0.1) is False
    assert L.impute_entry(L.BoolDomain(), L.QuadLoss(), 0.6) is True and L.impute_entry(L.BoolDomain(), L.QuadLoss(), 0.4) is False
    assert L.impute_entry(L.CountDomain(10), L.PoissonLoss(10), math.log(3.4)) == 3
    assert L.impute_entry(L.CategoricalDomain(3), L.MultinomialLoss(3), [0.1, 2.0, -1.0]) == 2
    assert L.impute_entry(L.OrdinalDomain(1, 3), L.MultinomialOrdinalLoss(3), [-0.1, -3.0]) == 2   # p = [1-e^-.1, e^-.1 - e^-3, e^-3]
    assert L.impute_entry(L.OrdinalDomain(1, 4), L.BvSLoss(4), [2.0, 1.0, -1.0]) == 3               # two thresholds passed
    assert L.error_metric_entry(L.PeriodicDomain(2.0), L.PeriodicLoss(2.0), 0.5, 2.5) == pytest.approx(0.0, abs=1e-30)
    assert L.error_metric_entry(L.BoolDomain(), L.LogisticLoss(), 0.3, True) == 0.0 and L.error_metric_entry(L.BoolDomain(), L.LogisticLoss(), 0.3, False) == 1.0
    with pytest.raises(ValueError):
        L.impute_entry(L.RealDomain(), L.LogisticLoss(), 0.3)


@pytest.mark.parametrize("loss", SCALAR + VECTOR, ids=lambda l: repr(l))
def test_oracle_impute_matches_python_mirror(loss):
    rng = np.random.default_rng(3)
    for dom in DOMAINS + [L.default_domain(loss)]:
        for _ in range(25):
            u = rng.standard_normal(loss.embedding_dim) * 2.0 if loss.embedding_dim > 1 else float(rng.standard_normal() * 3)
            try:
                ref = L.impute_entry(dom, loss, u)
            except (TypeError, ValueError):
                with pytest.raises(TypeError):
                    O.impute_entry(dom, loss, u)
                break
            assert O.impute_entry(dom, loss, u) == pytest.approx(float(ref), rel=1e-13)


def heterogeneous_model(rng, m=60):
    losses = [L.QuadLoss(), L.L1Loss(), L.HuberLoss(), L.PeriodicLoss(1), L.OrdinalHingeLoss(1, 10), L.LogisticLoss(), L.WeightedHingeLoss(),
              L.MultinomialLoss(4), L.BvSLoss(5), L.MultinomialOrdinalLoss(4), L.PoissonLoss(30), L.OvALoss(3)]
    k = 4
    d = L.embedding_dim(losses)
    X, Y = rng.standard_normal((k, m)), rng.standard_normal((k, d))
    return losses, X, Y, k


def test_imputation_is_consistent():
    """test/err_test.jl:33-51: A = impute(doms, losses, X'Y) has zero error metric at (X, Y), standardized or not."""
    rng = np.random.default_rng(4)
    losses, X, Y, k = heterogeneous_model(rng)
    api = O.oracle_api()
    m = X.shape[1]
    g0 = L.GLRM(np.ones((m, len(losses))), losses, L.ZeroReg(), L.ZeroReg(), k, X=X, Y=Y, checknan=False)
    A = L.impute(g0, engine=api)
    doms = [L.default_domain(l) for l in losses]
    U = X.T @ Y
    for f, (lo, (y0, y1)) in enumerate(zip(losses, L.get_yidxs(losses))):      # the matrix equals the entry-wise mirror
        for i in range(0, m, 7):
            assert A[i, f] == pytest.approx(float(L.impute_entry(doms[f], lo, U[i, y0:y1] if y1 - y0 > 1 else U[i, y0])), rel=1e-12)
    g = L.GLRM(A, losses, L.ZeroReg(), L.ZeroReg(), k, X=X, Y=Y)
    assert L.error_metric(g, engine=api) == 0.0 and L.error_metric(g, standardize=True, engine=api) == 0.0
    Xp = X + 0.3 * rng.standard_normal(X.shape)
    e_raw, e_std = L.error_metric(g, Xp, Y, engine=api), L.error_metric(g, Xp, Y, standardize=True, engine=api)
    assert e_raw > 0 and e_std > 0 and e_raw != e_std
    # transcription of raw / std error metric with the Python mirrors
    tot_raw = tot_std = 0.0
    Up = Xp.T @ Y
    for f, (lo, (y0, y1)) in enumerate(zip(losses, L.get_yidxs(losses))):
        errs = [L.error_metric_entry(doms[f], lo, Up[i, y0:y1] if y1 - y0 > 1 else Up[i, y0], A[i, f]) for i in range(m)]
        cm = float(np.mean(A[:, f] ** 2))
        tot_raw += sum(errs)
        tot_std += sum(errs) / cm if cm != 0 else sum(errs)
    assert e_raw == pytest.approx(tot_raw, rel=1e-12) and e_std == pytest.approx(tot_std, rel=1e-12)


def test_error_metric_as_cross_validation_error_fn():
    """test/err_test.jl:26,57: error_fn = error_metric(glrm, X, Y, doms, standardize=true) inside cross_validate."""
    rng = np.random.default_rng(5)
    losses, X, Y, k = heterogeneous_model(rng, 50)
    api = O.oracle_api()
    g0 = L.GLRM(np.ones((50, len(losses))), losses, L.ZeroReg(), L.ZeroReg(), k, X=X, Y=Y)
    A = L.impute(g0, engine=api)
    g = L.GLRM(A, losses, L.QuadReg(0.05), L.QuadReg(0.05), k, rng=rng)
    doms = [L.default_domain(l) for l in losses]
    fn = lambda glrm, Xf, Yf, **kw: L.error_metric(glrm, Xf, Yf, doms, standardize=True, **kw)
    tr, te, _, _ = L.cross_validate(g, nfolds=3, params=L.ProxGradParams(max_iter=30), verbose=False, error_fn=fn, rng=rng, engine=api)
    assert np.all(np.isfinite(tr)) and np.all(te >= tr * 0.5)


def test_impute_missing_keeps_observed_entries_and_unsupported_pairs_fail():
    rng = np.random.default_rng(6)
    A = rng.standard_normal((20, 6))
    I, J = np.nonzero(rng.random((20, 6)) < 0.5)
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(), L.QuadReg(), 2, obs=(I, J), rng=rng)
    api = O.oracle_api()
    Ahat = L.impute_missing(g, engine=api)
    assert np.array_equal(Ahat[I, J], A[I, J]) and np.allclose(Ahat[0, :][~np.isin(np.arange(6), J[I == 0])], (g.X.T @ g.Y)[0, :][~np.isin(np.arange(6), J[I == 0])])
    gl = L.GLRM(A > 0, L.LogisticLoss(), L.QuadReg(), L.QuadReg(), 2, rng=rng)
    with pytest.raises(L.GLRMError):
        L.error_metric(gl, domains=[L.RealDomain()] * 6, engine=api)  # RealDomain + LogisticLoss: the reference errors out
