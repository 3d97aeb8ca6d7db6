"""Pin the oracle (and the host-side operator mirror) to every closed-form known answer the
reference holds for this path:
  test/quantitative_tests/test_loss.jl:10-26   LogisticLoss values, Int label coercion, 3*l
  test/quantitative_tests/test_loss.jl:34-50   HingeLoss values and grads
  examples/LowRankModelsDemo-v1.1.0.ipynb:95,115,182,202,222,242  QuadLoss / NonNeg / QuadReg values
  src/losses.jl:258-292 hand-derived OrdinalHinge answers
"""
import math

import numpy as np
import pytest

import lowrankmodels.jl_amd as L
import oracle as O

TOL = 1e-3  # the reference's own tolerance (test_loss.jl:4)
L1v, L0v = 1.31326168, 0.3132616875


def both(loss, u, a):
    """oracle value and host-mirror value must agree to rounding; return the oracle's."""
    av = 1.0 if a is True else (0.0 if a is False else a)
    if loss.classification:
        av = 1.0 if L.losses.myBool(a) else 0.0
    o = O.loss_evaluate(loss, u, av)
    h = loss.evaluate(u, a)
    assert o == pytest.approx(h, rel=1e-14, abs=1e-14)
    return o


def both_grad(loss, u, a):
    av = (1.0 if L.losses.myBool(a) else 0.0) if loss.classification else a
    o = O.loss_grad(loss, u, av)
    assert o == pytest.approx(loss.grad(u, a), rel=1e-14, abs=1e-14)
    return o


def test_logistic_kat():
    l = L.LogisticLoss()
    for u, a, want in [(1, True, L0v), (1, False, L1v), (-1, True, L1v), (-1, False, L0v),
                       (1, 1, L0v), (1, -1, L1v), (1, 0, L1v), (-1, 1, L1v), (-1, -1, L0v), (-1, 0, L0v)]:
        assert both(l, u, a) == pytest.approx(want, abs=TOL)
    assert both(3 * l, 1, False) == pytest.approx(3 * L1v, abs=TOL)
    # tighter: SURVEY.md Appendix B quotes the full doubles
    assert both(l, 1, True) == pytest.approx(0.31326168751822286, rel=1e-15)
    assert both(l, 1, False) == pytest.approx(1.3132616875182228, rel=1e-15)
    with pytest.raises(ValueError):
        l.evaluate(1, 2)  # myBool: InexactError


def test_hinge_kat():
    l = L.HingeLoss()
    for u, a, want in [(1, True, 0), (1, False, 2), (-1, True, 2), (-1, False, 0), (1, 1, 0), (1, -1, 2),
                       (1, 0, 2), (-1, 1, 2), (-1, -1, 0), (-1, 0, 0)]:
        assert both(l, u, a) == pytest.approx(want, abs=TOL)
    assert both(3 * l, 1, False) == pytest.approx(6, abs=TOL)
    assert both_grad(l, -1, True) == pytest.approx(-1, abs=TOL)
    assert both_grad(l, 2, True) == pytest.approx(0, abs=TOL)
    assert both_grad(l, -2, False) == pytest.approx(0, abs=TOL)
    assert both_grad(l, 2, False) == pytest.approx(1, abs=TOL)


def test_notebook_kat():
    assert both(L.QuadLoss(), 2.0, 4.0) == 4.0
    assert both(L.QuadLoss(), 1.0, 2.0) == 1.0
    for reg, x, want in [(L.NonNegConstraint(), [1.0], 0.0), (L.NonNegConstraint(), [-1.0], math.inf),
                         (L.QuadReg(5), [1.0], 5.0), (L.QuadReg(5), [-2.0], 20.0)]:
        assert O.reg_evaluate(reg, x) == want
        assert reg.evaluate(x) == want


def test_ordinal_hinge_kat():
    l = L.OrdinalHingeLoss(1, 10, 1.0)
    assert both(l, 3.4, 2) == pytest.approx(1.8, rel=1e-14)
    assert both_grad(l, 3.4, 2) == 2
    # integer and half-integer u: evaluate and grad use different floor/ceil conventions (A.4)
    for u in [0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 5.0, 8.5, 9.0, 9.5, 10.0, 12.0, -3.0]:
        for a in [1, 2, 5, 10]:
            both(l, u, a)
            both_grad(l, u, a)
    assert L.OrdinalHingeLoss().max == 10 and L.OrdinalHingeLoss(7).max == 7 and L.OrdinalHingeLoss(7).min == 1


ALL_LOSSES = [L.QuadLoss(0.7), L.L1Loss(1.3), L.HuberLoss(0.9, crossover=0.6), L.QuantileLoss(1.1, quantile=0.3),
              L.PeriodicLoss(2.5, 0.8), L.PoissonLoss(), L.OrdinalHingeLoss(1, 5, 1.2), L.LogisticLoss(0.5),
              L.WeightedHingeLoss(1.5, case_weight_ratio=2.0)]


@pytest.mark.parametrize("loss", ALL_LOSSES, ids=lambda l: type(l).__name__)
def test_oracle_matches_host_mirror_and_gradient(loss):
    rng = np.random.default_rng(3)
    for _ in range(200):
        u = float(rng.normal() * 2)
        if loss.classification:
            a = bool(rng.integers(0, 2))
        elif isinstance(loss, L.OrdinalHingeLoss):
            a = int(rng.integers(1, 6))
        elif isinstance(loss, L.PoissonLoss):
            a = int(rng.integers(0, 6))
        else:
            a = float(rng.normal())
        both(loss, u, a)
        g = both_grad(loss, u, a)
        # grad is the derivative of evaluate wherever the loss is smooth (Quad/Periodic/Poisson/Logistic)
        if isinstance(loss, (L.QuadLoss, L.PeriodicLoss, L.PoissonLoss, L.LogisticLoss)):
            h = 1e-6
            av = (1.0 if a else 0.0) if loss.classification else a
            fd = (O.loss_evaluate(loss, u + h, av) - O.loss_evaluate(loss, u - h, av)) / (2 * h)
            assert g == pytest.approx(fd, rel=1e-5, abs=1e-6)


def test_huber_grad_is_not_twice():
    # src/losses.jl:177: (u-a)*scale in the quadratic zone -- NOT 2(u-a)*scale; reproduced, not "fixed"
    l = L.HuberLoss(1.0, crossover=1.0)
    assert O.loss_grad(l, 0.25, 0.0) == 0.25


ALL_REGS = [L.ZeroReg(), L.QuadReg(0.3), L.OneReg(0.4), L.NonNegConstraint(), L.UnitOneSparseConstraint()]


@pytest.mark.parametrize("reg", ALL_REGS, ids=lambda r: type(r).__name__)
def test_regularizer_prox_and_evaluate(reg):
    rng = np.random.default_rng(5)
    for _ in range(50):
        k = int(rng.integers(1, 9))
        u = rng.normal(size=k)
        alpha = float(rng.random() + 0.01)
        p = O.reg_prox(reg, u, alpha)
        np.testing.assert_allclose(p, reg.prox(u, alpha), rtol=1e-15, atol=0)
        assert O.reg_evaluate(reg, u) == pytest.approx(reg.evaluate(u), rel=1e-14)
        assert O.reg_evaluate(reg, p) == pytest.approx(reg.evaluate(p), rel=1e-14)
        # prox lands in the domain of the regularizer
        assert math.isfinite(O.reg_evaluate(reg, p))
        # prox minimises alpha*r(x) + 1/2|x-u|^2 among a few random competitors
        if not isinstance(reg, L.UnitOneSparseConstraint):
            best = alpha * O.reg_evaluate(reg, p) + 0.5 * np.sum((p - u) ** 2)
            for _ in range(10):
                z = p + 0.1 * rng.normal(size=k)
                val = alpha * O.reg_evaluate(reg, z) + 0.5 * np.sum((z - u) ** 2)
                assert val >= best - 1e-12


def test_unit_one_sparse_semantics():
    r = L.UnitOneSparseConstraint()
    np.testing.assert_array_equal(O.reg_prox(r, [0.2, 0.9, 0.9, -1.0], 0.5), [0, 1, 0, 0])  # first maximal index
    assert O.reg_evaluate(r, [0, 0, 0]) == 0.0  # the all-zero vector evaluates to 0 (regularizers.jl:300-316)
    assert O.reg_evaluate(r, [0, 1, 0]) == 0.0
    assert O.reg_evaluate(r, [1, 1, 0]) == math.inf
    assert O.reg_evaluate(r, [0, 0.5, 0]) == math.inf


def test_scale_semantics():
    # `*` on a loss SETS the scale (src/losses.jl:63-64); on a regularizer it multiplies (regularizers.jl:40)
    assert (3 * L.QuadLoss(2.0)).scale == 3.0
    assert (3 * L.QuadReg(2.0)).scale == 6.0
    z = L.ZeroReg()
    assert z.mul_(5) is z and z.scale == 1.0  # mul!(::ZeroReg, _) is a no-op
