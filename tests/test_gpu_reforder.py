"""-m gpu: glrm_options.sum_order = 1 -- the engine's REFERENCE-ORDER validation sweeps (csrc/glrm_reforder.hip; SURVEY.md section 8(b)
`line_search_sum_order`): one lane per segment, list order, one accumulator per sum, Julia's pairwise reduce(+) for the column loss sums
of DiffLoss / ClassificationLoss columns and for sum(obj_by_col).  Against the oracle in its default (reference) order:
  * losses both sides evaluate with the same instructions (Quad, L1, Huber, Quantile, OrdinalHinge, WeightedHinge) with every
    regularizer: X, Y, objective[1:], trial and accept counts are IDENTICAL, bit for bit -- nothing is left of "summation order";
  * losses with exp / log / sin (Logistic, Poisson, Periodic: in-kernel routines vs libm): 1e-9;
  * columns beyond 1024 observations walk the pairwise tree; duplicates, unsorted lists, empty segments, per-row regularizers;
  * what the mode does not cover is refused, not approximated.
Reference: src/algorithms/proxgrad.jl:107-217, src/evaluate_fit.jl:24-55, src/losses.jl:623-638."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi

pytestmark = pytest.mark.gpu


def hip():
    return _capi.hip_api()


def both(pa, X0, Y0, params):
    O.set_threads(4)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, params)
    o_g, X_g, Y_g, st_g = cases.run_engine(hip(), pa, X0, Y0, params, sum_order=1)
    assert st_g["tiled"] == 128, st_g["tiled"]
    return (o_c, X_c, Y_c, st_c), (o_g, X_g, Y_g, st_g)


def identical(c, g):
    (o_c, X_c, Y_c, st_c), (o_g, X_g, Y_g, st_g) = c, g
    assert len(o_g) == len(o_c)
    assert np.array_equal(o_g[1:], o_c[1:]), np.max(np.abs(o_g[1:] - o_c[1:]) / np.abs(o_c[1:]))
    assert cases.rel_err(o_g[:1], o_c[:1]) < 1e-12      # the initial objective is summed per column first on the device
    assert np.array_equal(X_g, X_c) and np.array_equal(Y_g, Y_c)
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert st_g[key] == st_c[key], key


@pytest.mark.parametrize("name", ["c1", "c4", "nnmf", "kmeans"])
def test_golden_cases_bit_for_bit(name):
    """QuadLoss with QuadReg / NonNegConstraint (Inf start) / UnitOneSparse + inner iterations: the golden fixtures' models."""
    kwargs, params = cases.build_golden_case(name)
    pa = L.GLRM(**kwargs).problem_arrays()
    identical(*both(pa, np.asfortranarray(kwargs["X"]), np.asfortranarray(kwargs["Y"]), params))


@pytest.mark.parametrize("k", [5, 20, 64])
def test_c4_recipe_with_long_columns_walks_the_pairwise_tree(k):
    """3000 x 60, 30 observations per row: columns of 1500 observations -> reduce(+) splits at first + (last - first) >> 1 (blocks of 1024)."""
    m, n, q = 3000, 60, 30
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=1)
    assert (np.diff(colptr) > 1024).all()
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(3, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    X0, Y0 = np.asfortranarray(np.abs(X0) / k ** 0.5), np.asfortranarray(np.abs(Y0) / k ** 0.5)
    c, g = both(pa, X0, Y0, L.ProxGradParams(8.0, max_iter=25))       # stepsize 8: first trials are rejected, the trial passes walk the tree too
    identical(c, g)
    assert c[3]["trials_y"] > c[3]["accepts_y"]          # some column rejected a trial: the trial passes ran too


def mixed_model(rng, m, n, k, kinds, regs):
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) / k ** 0.5
    losses = []
    for j in range(n):
        kind = kinds[j % len(kinds)]
        if kind == "quad":
            losses.append(L.QuadLoss(float(rng.uniform(0.5, 2))))
        elif kind == "l1":
            losses.append(L.L1Loss(float(rng.uniform(0.5, 2))))
        elif kind == "huber":
            losses.append(L.HuberLoss(float(rng.uniform(0.5, 2)), crossover=0.7))
        elif kind == "quantile":
            losses.append(L.QuantileLoss(1.0, quantile=0.3))
        elif kind == "ordinal":
            A[:, j] = np.clip(np.round(3 + 1.5 * A[:, j]), 1, 5)
            losses.append(L.OrdinalHingeLoss(1, 5))
        elif kind == "hinge":
            A[:, j] = (A[:, j] > 0).astype(float)
            losses.append(L.WeightedHingeLoss(1.0, case_weight_ratio=2.0))
        elif kind == "logistic":
            A[:, j] = (A[:, j] > 0).astype(float)
            losses.append(L.LogisticLoss())
        elif kind == "poisson":
            A[:, j] = rng.poisson(np.exp(np.clip(A[:, j], -2, 2)))
            losses.append(L.PoissonLoss())
        elif kind == "periodic":
            losses.append(L.PeriodicLoss(2 * np.pi))
    nobs = int(m * n * 0.4)
    obs = (rng.integers(0, m, nobs), rng.integers(0, n, nobs))        # sampled with replacement: duplicates, unsorted lists, empty rows
    rx = [regs[i % len(regs)]() for i in range(m)]
    return L.GLRM(A, losses, rx, L.QuadReg(0.2), k, obs=obs, X=rng.standard_normal((k, m)) / k ** 0.25, Y=rng.standard_normal((k, n)) / k ** 0.25)


def test_exactly_evaluated_losses_with_every_regularizer_bit_for_bit():
    rng = np.random.default_rng(42)
    regs = [lambda: L.QuadReg(0.3), lambda: L.OneReg(0.05), L.NonNegConstraint, L.ZeroReg, L.UnitOneSparseConstraint]
    g = mixed_model(rng, 400, 60, 7, ["quad", "l1", "huber", "quantile", "ordinal", "hinge"], regs)
    pa = g.problem_arrays()
    identical(*both(pa, np.asfortranarray(g.X), np.asfortranarray(g.Y), L.ProxGradParams(max_iter=12, inner_iter=2)))


def test_transcendental_losses_agree_to_rounding():
    rng = np.random.default_rng(43)
    g = mixed_model(rng, 300, 45, 6, ["logistic", "poisson", "periodic", "quad"], [lambda: L.QuadReg(0.3)])
    pa = g.problem_arrays()
    (o_c, X_c, Y_c, st_c), (o_g, X_g, Y_g, st_g) = both(pa, np.asfortranarray(g.X), np.asfortranarray(g.Y), L.ProxGradParams(max_iter=6))
    assert len(o_g) == len(o_c) and cases.rel_err(o_g, o_c) < 1e-9
    assert cases.fro_err(X_g, X_c) < 1e-9 and cases.fro_err(Y_g, Y_c) < 1e-9


def test_the_mode_is_reported_sharded_and_refused_where_it_does_not_apply():
    import torch
    kwargs, params = cases.build_golden_case("mixed")
    pa = L.GLRM(**kwargs).problem_arrays()
    api = hip()
    h = api.create(pa, sum_order=1)
    try:
        for w in (0, 1):
            o = api.sum_order(h, w).asdict()
            assert o["family_name"] == "reference" and o["lanes"] == 1
        with pytest.raises(_capi.GLRMError) as ei:       # the sparse solver's fixed-step sweeps are not restated in this mode
            api.fit_sparse(h, L.SparseProxGradParams(max_iter=2), np.asfortranarray(kwargs["X"]), np.asfortranarray(kwargs["Y"]))
        assert ei.value.code == _capi.ERR_UNSUPPORTED
    finally:
        api.destroy(h)
    # two ragged shards on one device == the single handle (the mode has no family choice to disagree on)
    X0, Y0 = np.asfortranarray(kwargs["X"]), np.asfortranarray(kwargs["Y"])
    o1, X1, Y1, _ = cases.run_engine(api, pa, X0, Y0, params, sum_order=1)
    ps = L.ProxGradParams(params.stepsize, max_iter=len(o1) - 1, abs_tol=0.0, rel_tol=-1.0)
    o2, X2, Y2, _ = cases.run_shards_on_one_device(api, pa, X0, Y0, ps, [0, 17, pa.m], [0, 11, pa.n], x_chunks=2, sum_order=1)
    assert np.array_equal(o2, o1[1:]) and np.array_equal(X2, X1) and np.array_equal(Y2, Y1)
    # refused: multi-dimensional losses, the dense hand-over
    kw2, _ = cases.build_golden_case("mnl_ordinal")
    with pytest.raises(_capi.GLRMError) as ei:
        api.create(L.GLRM(**kw2).problem_arrays(), sum_order=1)
    assert ei.value.code == _capi.ERR_UNSUPPORTED
    rng = np.random.default_rng(1)
    gd = L.GLRM(rng.standard_normal((64, 48)), L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), 16)
    with pytest.raises(_capi.GLRMError) as ei:
        api.create(gd.problem_arrays(dense=True), sum_order=1)
    assert ei.value.code == _capi.ERR_UNSUPPORTED
    with pytest.raises(_capi.GLRMError):
        api.create(pa, sum_order=2)
