"""M-estimators, avgerror and the scale=true rewrite (src/losses.jl:116-352, src/modify_glrm.jl:31-58) -- closed forms checked by
hand and against a brute-force minimisation of sum_i l(u, a_i)."""
import math

import numpy as np
import pytest

import extras as E
import lowrankmodels.jl_amd as L
import oracle as O


def test_closed_forms_by_hand():
    a = np.array([1.0, 2.0, 3.0, 6.0])
    assert E.M_estimator(L.QuadLoss(), a) == 3.0 and E.avgerror(L.QuadLoss(), a) == pytest.approx(3.5)
    assert E.M_estimator(L.L1Loss(), a) == 2.5 and E.avgerror(L.L1Loss(2.0), a) == pytest.approx(2.0 * (1.5 + 0.5 + 0.5 + 3.5) / 4)
    assert E.M_estimator(L.QuantileLoss(quantile=0.25), a) == pytest.approx(1.75)            # Julia quantile([1,2,3,6], .25) = 1.75
    assert E.M_estimator(L.PoissonLoss(10), a) == pytest.approx(math.log(3.0))
    b = np.array([1.0, 1.0, 1.0, 0.0])
    assert E.M_estimator(L.LogisticLoss(), b) == pytest.approx(math.log(7) - math.log(1))   # log(N+d) - log(N-d)
    assert E.M_estimator(L.WeightedHingeLoss(1.0, case_weight_ratio=1.0), b) == 1.0          # r = 4/3 - 1 < 1
    assert E.M_estimator(L.WeightedHingeLoss(1.0, case_weight_ratio=1.0), 1 - b) == -1.0     # r = 3 > 1
    assert E.M_estimator(L.WeightedHingeLoss(1.0, case_weight_ratio=1.0), np.array([1.0, 0.0])) == 0.0
    t = np.array([0.1, 0.2, 0.3])
    m = E.M_estimator(L.PeriodicLoss(1.0), t)
    assert m == pytest.approx((1 / (2 * math.pi)) * math.atan(np.sum(np.sin(2 * math.pi * t)) / np.sum(np.cos(2 * math.pi * t))) + 0.5)


# (LogisticLoss is absent on purpose: the reference's log(N+d) - log(N-d) is not the minimiser, log(d/(N-d)) is; the formula is kept.)
@pytest.mark.parametrize("loss", [L.QuadLoss(2.0), L.L1Loss(), L.QuantileLoss(quantile=0.3), L.PoissonLoss(30)],
                         ids=lambda l: type(l).__name__)
def test_m_estimators_minimise_the_summed_loss(loss):
    rng = np.random.default_rng(1)
    a = rng.random(51) < 0.3 if loss.classification else (np.round(np.exp(rng.standard_normal(51))) if isinstance(loss, L.PoissonLoss) else rng.standard_normal(51))
    a = np.asarray(a, dtype=float)
    m = E.M_estimator(loss, a)
    f = lambda u: sum(loss.evaluate(u, ai) for ai in a)
    grid = m + np.linspace(-0.5, 0.5, 201)
    assert f(m) <= min(f(u) for u in grid) + 1e-9 * abs(f(m))


def test_scale_true_rewrites_the_model_like_equilibrate_variance():
    rng = np.random.default_rng(2)
    m, n = 40, 4
    A = np.column_stack([rng.standard_normal(m) * 3 + 1, rng.random(m) < 0.3, np.round(np.clip(3 + rng.standard_normal(m), 1, 5)), rng.standard_normal(m)])
    losses = [L.QuadLoss(), L.LogisticLoss(), L.OrdinalHingeLoss(1, 5), L.HuberLoss(2.0)]
    I, J = np.nonzero(rng.random((m, n)) < 0.8)
    g = L.GLRM(A, losses, L.QuadReg(0.5), [L.QuadReg(0.5), L.OneReg(0.2), L.ZeroReg(), L.QuadReg(1.0)], 3, obs=(I, J), scale=E.equilibrate_variance_, offset=True, rng=rng)
    for f, (l0, r0) in enumerate(zip(losses, [0.5, 0.2, 1.0, 1.0])):
        col = A[I[J == f], f].astype(float)
        assert g.losses[f].scale == pytest.approx(l0.scale / E.avgerror(l0, col))         # mul!(l, scale(l) / varlossi)
        base = g.ry[f].r
        if not isinstance(base, L.ZeroReg):
            assert base.scale == pytest.approx(r0 / np.var(col, ddof=1))                    # mul!(ry, scale(ry) / var(nomissing))
    assert all(isinstance(r, L.lastentry1) for r in g.rx) and isinstance(g.ry[0], L.lastentry_unpenalized)  # offset applied after scaling
    X, Y, ch = L.fit_b(g, L.ProxGradParams(max_iter=20), verbose=False, engine=O.oracle_api())
    assert np.isfinite(ch.objective[-1]) and ch.objective[-1] < ch.objective[1]
    with pytest.raises(NotImplementedError):
        L.GLRM(np.ones((5, 1)), L.MultinomialLoss(3), L.QuadReg(), L.QuadReg(), 2, scale=E.equilibrate_variance_)


def test_prob_scale():
    rng = np.random.default_rng(3)
    A = np.column_stack([rng.standard_normal(30) * 2, rng.standard_normal(30), rng.random(30) < 0.5])
    g = L.GLRM(A, [L.QuadLoss(), L.HuberLoss(), L.LogisticLoss(3.0)], L.QuadReg(), L.QuadReg(), 2)
    E.prob_scale_(g)
    assert g.losses[0].scale == pytest.approx(1 / (2 * np.var(A[:, 0], ddof=1)))
    assert g.losses[1].scale == pytest.approx(1 / (2 * E.avgerror(L.HuberLoss(), A[:, 1])))
    assert g.losses[2].scale == 1.0   # mul!(l, 1): `*`/mul! SET the scale (src/losses.jl:61-64)


def test_prob_scale_small_variance_column_is_rescaled():
    """TOL of prob_scale! is the module constant 1e-12 (src/regularizers.jl:25), not the 1e-3 keyword defaults of the MNL ordinal
    rules: a real column in small units (variance ~1e-4) is rescaled."""
    rng = np.random.default_rng(4)
    A = np.column_stack([0.01 * rng.standard_normal(40), rng.standard_normal(40)])
    v = np.var(A[:, 0], ddof=1)
    assert 1e-12 < v < 1e-3
    g = L.GLRM(A, [L.QuadLoss(), L.QuadLoss()], L.QuadReg(), L.QuadReg(), 2)
    E.prob_scale_(g)
    assert g.losses[0].scale == pytest.approx(1 / (2 * v)) and g.losses[0].scale > 1000
