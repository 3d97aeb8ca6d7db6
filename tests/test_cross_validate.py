"""cross_validate / cv_by_iter / regularization_path / get_train_and_test (src/cross_validate.jl) on the CPU side: the fold
models equal what the reference builds (sort_observations of the fold's obs), the fused path (subset of the resident parent
handle) equals building every fold from host arrays bit for bit, and the drivers return what a direct transcription of the
reference driver computes."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import crossval as CV


def model(rng, m=40, n=25, k=3, density=0.6, losses=None):
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) + 0.1 * rng.standard_normal((m, n))
    I, J = np.nonzero(rng.random((m, n)) < density)
    return L.GLRM(A, L.QuadLoss() if losses is None else losses, L.QuadReg(0.1), L.QuadReg(0.1), k, obs=(I, J),
                  X=rng.standard_normal((k, m)), Y=rng.standard_normal((k, n)))


def test_fold_models_are_sort_observations_of_the_fold():
    rng = np.random.default_rng(0)
    g = model(rng)
    I, J = CV.flatten_observations(g.observed_features)
    assert np.array_equal(I, np.repeat(np.arange(g.m), np.diff(g._rowptr))) and np.array_equal(J, g._colidx)
    tags = CV.getfolds((I, J), 4, g.m, g.n, do_check=True, rng=rng)
    sp = CV._Split(g)
    assert sp.canonical
    for f in range(4):
        for keep in (tags != f, tags == f):
            c = sp.child(keep)
            ref = L.GLRM(g.A, g.losses, g.rx, g.ry, g.k, obs=(I[keep], J[keep]), X=g.X, Y=g.Y)  # GLRM(...; obs) -> sort_observations
            for name in ("_rowptr", "_colidx", "_rowvals", "_colptr", "_rowidx", "_colvals"):
                assert np.array_equal(getattr(c, name), getattr(ref, name)), name


def test_non_canonical_column_view_is_detected():
    kwargs, _ = cases.build_golden_case("mixed")  # observed_examples deliberately unrelated to observed_features
    g = L.GLRM(**kwargs)
    assert not CV._Split(g).canonical


def direct_cross_validate(g, tags, nfolds, params, api):
    """The reference driver written out with plain GLRM constructions and fresh handles per fold."""
    I, J = CV.flatten_observations(g.observed_features)
    tr_err, te_err = [], []
    for f in range(nfolds):
        tr = L.GLRM(g.A, g.losses, g.rx, g.ry, g.k, obs=(I[tags != f], J[tags != f]), X=g.X, Y=g.Y)
        te = L.GLRM(g.A, g.losses, g.rx, g.ry, g.k, obs=(I[tags == f], J[tags == f]), X=g.X, Y=g.Y)
        X, Y, ch = L.fit_b(tr, params, verbose=False, engine=api)
        tr_err.append(L.objective(tr, X, Y, include_regularization=False, engine=api) / int(tr._rowptr[-1]))
        te_err.append(L.objective(te, X, Y, include_regularization=False, engine=api) / int(te._rowptr[-1]))
    return np.array(tr_err), np.array(te_err)


@pytest.mark.parametrize("fused", [True, False])
def test_cross_validate_matches_direct_transcription(fused):
    rng = np.random.default_rng(1)
    g = model(rng)
    api = O.oracle_api()
    O.set_threads(2)
    tags = rng.integers(0, 3, int(g._rowptr[-1]))
    p = L.ProxGradParams(max_iter=15)
    tr, te, trg, teg = L.cross_validate(g, nfolds=3, params=p, verbose=False, groups=tags, engine=api, fused=fused)
    tr_d, te_d = direct_cross_validate(g, tags, 3, p, api)
    assert np.array_equal(tr, tr_d) and np.array_equal(te, te_d)  # same arithmetic on the same lists: bit-identical
    assert np.all(te > tr * 0.5) and len(trg) == 3 and trg[0].X.shape == g.X.shape
    assert not np.array_equal(trg[0].X, g.X) and np.array_equal(g.X, model(np.random.default_rng(1)).X)  # the parent is untouched


def test_use_folds_and_do_obs_check():
    rng = np.random.default_rng(2)
    g = model(rng, density=0.9)
    tr, te, trg, teg = L.cross_validate(g, nfolds=4, use_folds=2, params=L.ProxGradParams(max_iter=5), verbose=False, rng=rng,
                                        engine=O.oracle_api(), do_obs_check=True)
    assert np.isfinite(tr[:2]).all() and np.isnan(tr[2:]).all() and trg[2] is None
    sparse = model(rng, m=6, n=6, density=0.2)
    with pytest.raises(ValueError):
        L.cross_validate(sparse, nfolds=3, verbose=False, rng=rng, engine=O.oracle_api(), do_obs_check=True)


def test_regularization_path_reuses_one_handle_and_matches_fresh_fits():
    rng = np.random.default_rng(3)
    g = model(rng)
    api = O.oracle_api()
    draws = rng.random(int(g._rowptr[-1]))
    regs = [10.0, 1.0, 0.1]
    p = L.ProxGradParams(max_iter=12)
    tr, te, tt, rp = L.regularization_path(g, params=p, reg_params=regs, holdout_proportion=0.2, verbose=False, groups=draws, engine=api)
    assert list(rp) == regs and np.all(np.diff(tt) > 0)
    # transcription: split, then for each reg_param scale the regularizers, warm-start fit, evaluate (src/cross_validate.jl:211-240)
    I, J = CV.flatten_observations(g.observed_features)
    test = draws < 0.2
    trg = L.GLRM(g.A, g.losses, g.rx, g.ry, g.k, obs=(I[~test], J[~test]), X=g.X, Y=g.Y)
    teg = L.GLRM(g.A, g.losses, g.rx, g.ry, g.k, obs=(I[test], J[test]), X=g.X, Y=g.Y)
    for i, r in enumerate(regs):
        L.scale_regularizer_(trg, r)
        trg.close()  # fresh handle per fit: the path above must give the same numbers from ONE handle + set_regularizers
        X, Y, _ = L.fit_b(trg, p, verbose=False, engine=api)
        assert tr[i] == L.objective(trg, X, Y, include_regularization=False, engine=api) / int(trg._rowptr[-1])
        assert te[i] == L.objective(teg, X, Y, include_regularization=False, engine=api) / int(teg._rowptr[-1])
    assert te[0] > te[-1]  # the heavily regularized model fits worst


def test_cv_by_iter_equals_one_long_fit():
    rng = np.random.default_rng(4)
    g = model(rng)
    api = O.oracle_api()
    draws = rng.random(int(g._rowptr[-1]))
    p = L.ProxGradParams(1.0, max_iter=8, abs_tol=0.0, rel_tol=0.0)
    tr, te = L.cv_by_iter(g, 0.15, p, verbose=False, groups=draws, engine=api)
    train, test = L.get_train_and_test(g, 0.15, groups=draws, engine=api)
    X, Y, ch = L.fit_b(train, L.ProxGradParams(1.0, max_iter=8, abs_tol=0.0, rel_tol=0.0), verbose=False, engine=api)
    # eight warm-started one-iteration fits restart the step sizes each time: only the first iteration coincides
    assert tr[0] == ch.objective[1] and len(tr) == 8 and tr[-1] < tr[0] and te[-1] < te[0]


def test_subset_abi_keeps_order_and_duplicates():
    kwargs, _ = cases.build_golden_case("mixed")  # duplicates in both views
    g = L.GLRM(**kwargs)
    api = O.oracle_api()
    rng = np.random.default_rng(5)
    rt, ct = rng.integers(0, 3, len(g._colidx)).astype(np.uint8), rng.integers(0, 3, len(g._rowidx)).astype(np.uint8)
    h = api.create(g.problem_arrays())
    try:
        for match, inv in ((1, False), (1, True), (7, False)):
            hc = api.subset(h, rt, ct, match, inv)
            try:
                st = api.kernel_stats(hc)
                kr, kc = (rt == match) != inv, (ct == match) != inv
                assert st["nnz_rows"] == kr.sum() and st["nnz_cols"] == kc.sum()
                # the child's objective equals the objective of a model built from the kept entries of each view
                feats = [list(g._colidx[g._rowptr[e]:g._rowptr[e + 1]][kr[g._rowptr[e]:g._rowptr[e + 1]]]) for e in range(g.m)]
                exs = [list(g._rowidx[g._colptr[f]:g._colptr[f + 1]][kc[g._colptr[f]:g._colptr[f + 1]]]) for f in range(g.n)]
                ref = L.GLRM(g.A, g.losses, g.rx, g.ry, g.k, observed_features=feats, observed_examples=exs, X=g.X, Y=g.Y)
                X, Y = np.asfortranarray(g.X), np.asfortranarray(g.Y)
                assert api.objective(hc, X, Y, True) == L.objective(ref, X, Y, engine=api)
            finally:
                api.destroy(hc)
    finally:
        api.destroy(h)
