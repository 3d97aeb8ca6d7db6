"""Multi-dimensional losses, block regularizers and offsets on the CPU side: known answers of the operators, the oracle's
C restatement against the Python mirrors (two independent transcriptions of src/losses.jl:360-620 and
src/regularizers.jl:163-189,356-411), the host-side data model (Y is k x embedding_dim) and the argument checks."""
import math

import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi

VLOSSES = [L.MultinomialLoss(4), L.MultinomialLoss(6, .5), L.OvALoss(3), L.OvALoss(5, 2.0, bin_loss=L.HingeLoss()),
           L.BvSLoss(4), L.BvSLoss(6, 0.3, bin_loss=L.HingeLoss(2.0)), L.OrdisticLoss(5), L.MultinomialOrdinalLoss(3),
           L.MultinomialOrdinalLoss(6, 1.7)]


def test_known_answers():
    # u = 0: every category equally likely
    assert O.vloss_evaluate(L.MultinomialLoss(3), [0, 0, 0], 2) == pytest.approx(math.log(3), rel=1e-15)
    assert O.vloss_evaluate(L.OrdisticLoss(4), [0, 0, 0, 0], 1) == pytest.approx(math.log(4), rel=1e-15)
    assert O.vloss_evaluate(L.OvALoss(3), [0, 0, 0], 1) == pytest.approx(3 * math.log(2), rel=1e-15)
    assert O.vloss_evaluate(L.BvSLoss(4), [0, 0, 0], 4) == pytest.approx(3 * math.log(2), rel=1e-15)
    # multinomial is shift invariant and its gradient sums to 0 (softmax - e_a)
    g = O.vloss_grad(L.MultinomialLoss(4), [0.3, -1.0, 2.0, 0.1], 3)
    p = np.exp([0.3, -1.0, 2.0, 0.1]); p /= p.sum()
    np.testing.assert_allclose(g, p - np.eye(4)[2], rtol=1e-13)
    assert O.vloss_evaluate(L.MultinomialLoss(4), [0.3, -1.0, 2.0, 0.1], 3) == pytest.approx(-math.log(p[2]), rel=1e-13)
    # MultinomialOrdinalLoss: thresholds are forced negative and decreasing (enforce_MNLOrdRules, TOL = 1e-3)
    l = L.MultinomialOrdinalLoss(3)
    assert O.vloss_evaluate(l, [-1.0, -2.0], 1) == pytest.approx(-math.log(1 - math.exp(-1)), rel=1e-14)
    assert O.vloss_evaluate(l, [-1.0, -2.0], 3) == pytest.approx(2.0, rel=1e-15)
    assert O.vloss_evaluate(l, [-1.0, -2.0], 2) == pytest.approx(-math.log(math.exp(-1) - math.exp(-2)), rel=1e-14)
    assert O.vloss_evaluate(l, [5.0, 7.0], 3) == pytest.approx(2e-3, rel=1e-12)      # u -> [-1e-3, -2e-3]
    # hinge one-vs-all: margins
    assert O.vloss_evaluate(L.OvALoss(3, bin_loss=L.HingeLoss()), [2.0, -2.0, 0.5], 1) == pytest.approx(1.5)


@pytest.mark.parametrize("loss", VLOSSES, ids=lambda l: repr(l))
def test_vector_losses_oracle_vs_python_mirror_and_finite_differences(loss):
    rng = np.random.default_rng(5)
    d = loss.embedding_dim
    for _ in range(20):
        u = rng.standard_normal(d) * 1.5
        if isinstance(loss, L.MultinomialOrdinalLoss):
            u = -np.cumsum(0.2 + rng.random(d))  # strictly inside the feasible cone: the loss is smooth there
        for a in range(1, loss.max + 1):
            e_c, e_p = O.vloss_evaluate(loss, u, a), loss.evaluate(u, a)
            assert e_c == pytest.approx(e_p, rel=1e-13, abs=1e-15)
            g_c, g_p = O.vloss_grad(loss, u, a), loss.grad(u, a)
            np.testing.assert_allclose(g_c, g_p, rtol=1e-12, atol=1e-15)
            if getattr(loss, "bin_loss", None) is not None and not isinstance(loss.bin_loss, L.LogisticLoss):
                continue  # hinge: not differentiable at the kinks
            h = 1e-6
            fd = np.array([(O.vloss_evaluate(loss, u + h * np.eye(d)[j], a) - O.vloss_evaluate(loss, u - h * np.eye(d)[j], a)) / (2 * h)
                           for j in range(d)])
            np.testing.assert_allclose(g_c, fd, rtol=2e-6, atol=2e-8)


WRAPPED = [L.lastentry1(L.QuadReg(0.3)), L.lastentry1(L.OneReg(0.2)), L.lastentry1(L.NonNegConstraint()), L.lastentry1(L.ZeroReg()),
           L.lastentry1(L.UnitOneSparseConstraint()), L.lastentry_unpenalized(L.QuadReg(0.7)), L.lastentry_unpenalized(L.OneReg(0.1)),
           L.lastentry_unpenalized(L.NonNegConstraint()), L.OrdinalReg(L.QuadReg(0.4)), L.OrdinalReg(L.OneReg(0.3)),
           L.OrdinalReg(L.ZeroReg()), L.MNLOrdinalReg(L.QuadReg(0.05)), L.MNLOrdinalReg(L.NonNegConstraint())]


@pytest.mark.parametrize("reg", WRAPPED, ids=lambda r: repr(r))
def test_block_regularizers_oracle_vs_python_mirror(reg):
    rng = np.random.default_rng(9)
    for d in (1, 3, 6):
        if d > 1 and isinstance(reg, L.lastentry1):
            continue  # lastentry1 sits on X: vectors only
        for _ in range(10):
            k = int(rng.integers(2, 7))
            u = rng.standard_normal((k, d)) if d > 1 else rng.standard_normal(k)
            alpha = float(rng.random() + 0.05)
            p_c, p_p = O.reg_prox_block(reg, u, alpha), np.asarray(reg.prox(u, alpha))
            np.testing.assert_allclose(p_c, p_p, rtol=1e-13, atol=1e-15)
            for blk in (u, p_p):
                e_c, e_p = O.reg_evaluate_block(reg, blk), reg.evaluate(blk)
                assert (e_c == e_p) if not np.isfinite(e_p) else e_c == pytest.approx(e_p, rel=1e-13, abs=1e-300)


def test_wrapper_semantics():
    x = np.array([0.5, -2.0, 3.0])
    r = L.lastentry1(L.QuadReg(1.0))
    assert r.evaluate(x) == math.inf and O.reg_evaluate_block(r, x) == math.inf          # a[end] != 1
    px = r.prox(x, 0.5)
    assert px[-1] == 1 and np.allclose(px[:2], x[:2] / 2)                                 # prox of QuadReg: u / (1 + 2 alpha)
    assert r.evaluate(px) == pytest.approx(np.sum(px[:2] ** 2))
    ru = L.lastentry_unpenalized(L.QuadReg(1.0))
    assert ru.evaluate(x) == pytest.approx(0.25 + 4.0) and ru.prox(x, 0.5)[-1] == 3.0     # the last entry is free
    assert L.lastentry_unpenalized(L.OrdinalReg(L.QuadReg())).__class__ is L.OrdinalReg   # src/regularizers.jl:386
    blk = np.array([[1.0, 3.0], [0.5, -0.2]])
    po = L.OrdinalReg(L.ZeroReg()).prox(blk, 1.0)
    assert np.allclose(po[0], 2.0) and np.allclose(po[1], blk[1])                         # first k-1 rows: column mean
    pm = L.MNLOrdinalReg(L.ZeroReg()).prox(np.array([[1.0, 3.0, 0.0], [0.5, 0.7, -4.0]]), 1.0)
    assert np.allclose(pm[1], [-1e-3, -2e-3, -4.0])                                       # negative, decreasing last row
    r.mul_(3.0)
    assert r.r.scale == 3.0 and r.descriptor() == (L.QuadReg().kind, 1, 3.0)


def test_glrm_data_model_with_embedding():
    kwargs, _ = cases.build_multidim_case("categorical_mix")
    kwargs.pop("Y")
    g = L.GLRM(**kwargs)
    assert g.Y.shape == (3, L.embedding_dim(g.losses)) == (3, 4 + 1 + 3 + 4 + 4 + 1 + 2 + 3)
    assert L.get_yidxs(g.losses)[:3] == [(0, 4), (4, 5), (5, 8)]
    pa = g.problem_arrays()
    assert pa.d == g.Y.shape[1] and list(pa.ystart[:4]) == [0, 4, 5, 8]
    assert [int(x) for x in pa.losses["dim"][:4]] == [4, 0, 3, 4]
    with pytest.raises(ValueError):
        L.GLRM(kwargs["A"], kwargs["losses"], L.QuadReg(), L.QuadReg(), 3, Y=np.zeros((3, 8)))  # Y must be k x embedding_dim


def test_offset_constructor():
    """GLRM(...; offset=true) = add_offset!: rx -> lastentry1(rx), ry -> lastentry_unpenalized(ry) (src/modify_glrm.jl:20-25)."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((12, 7))
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.2), 3, offset=True)
    assert all(isinstance(r, L.lastentry1) for r in g.rx) and all(isinstance(r, L.lastentry_unpenalized) for r in g.ry)
    assert not g.dense_eligible()
    pa = g.problem_arrays()
    assert int(pa.rx["wrap"][0]) == 1 and int(pa.ry["wrap"][0]) == 2
    go = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), [L.OrdinalReg(L.QuadReg(0.2))] * 7, 3, offset=True)
    assert all(isinstance(r, L.OrdinalReg) for r in go.ry)  # no second offset on an ordinal block


def test_argument_checks():
    api = O.oracle_api()
    kwargs, _ = cases.build_multidim_case("categorical_mix")
    A = np.array(kwargs["A"])
    A[kwargs["observed_features"][0][0] * 0 + 0, 0] = 5.0  # MultinomialLoss(4) column: level 5 does not exist
    kwargs["observed_features"][0] = sorted(set(kwargs["observed_features"][0]) | {0})
    kwargs["A"] = A
    with pytest.raises(_capi.GLRMError) as ei:
        api.create(L.GLRM(**kwargs).problem_arrays())
    assert ei.value.code == _capi.ERR_NONFINITE and "level" in str(ei.value)
    kwargs, _ = cases.build_multidim_case("categorical_mix")
    pa = L.GLRM(**kwargs).problem_arrays()
    bad = pa.losses.copy(); bad["dim"][0] = 40
    pa.losses = bad
    with pytest.raises(_capi.GLRMError) as ei:
        api.create(pa)
    assert ei.value.code == _capi.ERR_INVALID
    pa = L.GLRM(**kwargs).problem_arrays()
    bad = pa.losses.copy(); bad["dim"][1] = 3          # a scalar loss cannot own 3 columns of Y
    pa.losses = bad
    with pytest.raises(_capi.GLRMError):
        api.create(pa)
    pa = L.GLRM(**kwargs).problem_arrays()
    bad = pa.ry.copy(); bad["wrap"][0] = 3
    pa.ry = bad
    with pytest.raises(_capi.GLRMError):
        api.create(pa)
    with pytest.raises(NotImplementedError):
        L.OvALoss(3, bin_loss=L.QuadLoss())
    with pytest.raises(ValueError):
        L.MultinomialLoss(64)                             # embedding dimension above GLRM_MAX_EMBEDDING_DIM


@pytest.mark.parametrize("name", list(cases.MULTIDIM_CASES))
def test_step_level_api_matches_whole_fit_and_shards(name):
    """The outer loop driven through the step-level entry points on two row/column shards reproduces glrm_cpu_fit
    bit for bit (the multi-GPU host relies on it), including the Y block spans of multi-dimensional columns."""
    kwargs, p = cases.build_multidim_case(name)
    p = L.ProxGradParams(max_iter=6, inner_iter_X=p.inner_iter_X, inner_iter_Y=p.inner_iter_Y)
    g = L.GLRM(**kwargs)
    api = O.oracle_api()
    O.set_threads(2)
    obj_ref, X_ref, Y_ref, _ = cases.run_engine(api, g.problem_arrays(), kwargs["X"], kwargs["Y"], p)
    rb, cb = [0, g.m // 2, g.m], [0, g.n // 3, g.n]
    hs = [api.create(g.problem_arrays(rows=(rb[r], rb[r + 1]), cols=(cb[r], cb[r + 1]))) for r in range(2)]
    try:
        ys = g.problem_arrays().ystart
        X, Y = np.array(kwargs["X"], order="F"), np.array(kwargs["Y"], order="F")
        for h in hs:
            api.set_factors(h, X, Y)
            api.reset_stepsizes(h, p.stepsize)

        def exchange(rows):
            Xs, Ys = [], []
            for h in hs:
                Xr, Yr = np.zeros_like(X), np.zeros_like(Y)
                api.get_factors(h, Xr, Yr)
                Xs.append(Xr); Ys.append(Yr)
            for r in range(2):
                if rows:
                    X[:, rb[r]:rb[r + 1]] = Xs[r][:, rb[r]:rb[r + 1]]
                else:
                    Y[:, ys[cb[r]]:ys[cb[r + 1]]] = Ys[r][:, ys[cb[r]]:ys[cb[r + 1]]]
            for h in hs:
                api.set_factors(h, X, Y)

        for it in range(p.max_iter):
            if p.inner_iter_X > 1 or p.inner_iter_Y > 1:
                for h in hs:
                    api.reset_stepsizes(h, p.stepsize)
            for _ in range(p.inner_iter_X):
                for h in hs:
                    api.step_x(h, p.min_stepsize)
            exchange(True)
            for _ in range(p.inner_iter_Y):
                for h in hs:
                    api.step_y(h, p.min_stepsize)
                exchange(False)
        assert np.array_equal(X, X_ref) and np.array_equal(Y, Y_ref)
    finally:
        for h in hs:
            api.destroy(h)


def test_sparse_solver_on_multidim_model():
    """fit!(glrm, SparseProxGradParams) (src/algorithms/sparse_proxgrad.jl) accepts the same models."""
    kwargs, _ = cases.build_multidim_case("categorical_mix")
    g = L.GLRM(**kwargs)
    api = O.oracle_api()
    h = api.create(g.problem_arrays())
    try:
        X, Y = np.array(kwargs["X"], order="F"), np.array(kwargs["Y"], order="F")
        obj, _ = api.fit_sparse(h, L.SparseProxGradParams(max_iter=25), X, Y)
        assert obj[-1] < obj[0] and np.all(np.diff(obj[:-1]) < 0)
        assert api.objective(h, X, Y) == pytest.approx(obj[-1], rel=1e-12)
    finally:
        api.destroy(h)
