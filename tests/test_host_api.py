"""Host-side mirror of the reference interface: GLRM construction, Omega bookkeeping (bit-exact),
params, fit!/fit semantics.  The engine used here is the CPU oracle (test hook); the HIP engine runs
the same host code in the -m gpu tests."""
import numpy as np
import pytest
import scipy.sparse as sp

import extras as E
import lowrankmodels.jl_amd as L
import oracle as O


def test_params_defaults_and_inner_iter_merge():
    p = L.ProxGradParams()
    assert (p.stepsize, p.max_iter, p.inner_iter_X, p.inner_iter_Y) == (1.0, 100, 1, 1)
    assert (p.abs_tol, p.rel_tol, p.min_stepsize) == (1e-5, 1e-4, 0.01)
    p = L.ProxGradParams(2, inner_iter=10, inner_iter_X=3)
    assert (p.inner_iter_X, p.inner_iter_Y, p.min_stepsize) == (10, 10, 0.02)  # max-merge, 0.01*stepsize
    assert isinstance(L.Params(1, max_iter=7), L.ProxGradParams) and L.Params(1, max_iter=7).max_iter == 7
    assert isinstance(L.HipProxGradParams(), L.AbstractParams)


def test_constructor_dimension_checks():
    A = np.zeros((4, 3))
    with pytest.raises(ValueError, match="as many losses"):
        L.GLRM(A, [L.QuadLoss()] * 2, L.ZeroReg(), L.ZeroReg(), 2)
    with pytest.raises(ValueError, match="X regularizer"):
        L.GLRM(A, L.QuadLoss(), [L.ZeroReg()] * 3, L.ZeroReg(), 2)
    with pytest.raises(ValueError, match="Y regularizer"):
        L.GLRM(A, L.QuadLoss(), L.ZeroReg(), [L.ZeroReg()] * 4, 2)
    with pytest.raises(ValueError, match="X must be of size"):
        L.GLRM(A, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), 2, X=np.zeros((3, 3)))
    with pytest.raises(ValueError, match="Y must be of size"):
        L.GLRM(A, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), 2, Y=np.zeros((2, 4)))
    g = L.GLRM(A, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), 2, X=np.ones((4, 2)))  # m x k is transposed (glrm.jl:57-60)
    assert g.X.shape == (2, 4)
    An = A.copy()
    An[1, 2] = np.nan
    with pytest.raises(ValueError, match=r"Observed value in entry \(1, 2\) is NaN"):
        L.GLRM(An, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), 2)
    L.GLRM(An, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), 2, obs=[(0, 0), (1, 1)])  # unobserved NaN is fine
    with pytest.raises(ValueError, match="not a Bool"):
        L.GLRM(np.full((4, 3), 2.0), L.LogisticLoss(), L.ZeroReg(), L.ZeroReg(), 2)


def test_sort_observations_keeps_order_and_duplicates():
    # src/modify_glrm.jl:5-18: push! in obs order, duplicates retained (test/hello_world.jl:48 samples with replacement)
    obs = [(2, 1), (0, 1), (2, 0), (2, 1), (1, 1), (0, 0)]
    rowptr, colidx, colptr, rowidx = L.sort_observations(obs, 3, 2)
    feats = [list(colidx[rowptr[e]:rowptr[e + 1]]) for e in range(3)]
    exs = [list(rowidx[colptr[f]:colptr[f + 1]]) for f in range(2)]
    assert feats == [[1, 0], [1], [1, 0, 1]]
    assert exs == [[2, 0], [2, 0, 2, 1]]
    I, J = np.array([o[0] for o in obs]), np.array([o[1] for o in obs])
    r2 = L.sort_observations((I, J), 3, 2)
    for a, b in zip((rowptr, colidx, colptr, rowidx), r2):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):
        L.sort_observations([(0, 0)], 2, 2, check_empty=True)


def test_sparse_pattern_bookkeeping():
    """test/sparse_test.jl:21-46: the observed sets are exactly the nonzero pattern."""
    rng = np.random.default_rng(0)
    m, n, k = 100, 100, 3
    A = sp.random(m, n, density=0.5, random_state=rng, format="csc")
    g = L.GLRM(A, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), k)
    assert len(g.observed_features) == m and len(g.observed_examples) == n
    D = A.toarray()
    for i in range(m):
        of = set(g.observed_features[i].tolist())
        for j in range(n):
            assert (j in of) == (D[i, j] != 0.0)
    for j in range(n):
        oe = set(g.observed_examples[j].tolist())
        for i in range(m):
            assert (i in oe) == (D[i, j] != 0.0)
    # findall is column-major => both lists ascending (src/glrm.jl:46-48)
    assert all(np.all(np.diff(f) > 0) for f in g.observed_features)
    assert all(np.all(np.diff(e) > 0) for e in g.observed_examples)
    # values gathered at the observed entries, both views
    pa = g.problem_arrays()
    rows = np.repeat(np.arange(m), np.diff(pa.rowptr))
    np.testing.assert_array_equal(pa.rowvals, D[rows, pa.colidx])
    cols = np.repeat(np.arange(n), np.diff(pa.colptr))
    np.testing.assert_array_equal(pa.colvals, D[pa.rowidx, cols])


def test_default_is_fully_observed_and_singletons_broadcast():
    A = np.arange(12.0).reshape(4, 3)
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), 2)
    assert [list(f) for f in g.observed_features] == [[0, 1, 2]] * 4
    assert [list(e) for e in g.observed_examples] == [[0, 1, 2, 3]] * 3
    assert len(g.losses) == 3 and len(g.rx) == 4 and len(g.ry) == 3
    pa = g.problem_arrays()
    assert len(pa.losses) == 1 and len(pa.rx) == 1 and len(pa.ry) == 1  # homogeneous lists collapse in the ABI
    g2 = L.GLRM(A, [L.QuadLoss(), L.HuberLoss(), L.QuadLoss()], [L.QuadReg(), L.OneReg(5), L.NonNegConstraint(), L.QuadReg()],
                L.QuadReg(), 2)
    pb = g2.problem_arrays(rows=(1, 3), cols=(0, 2))
    assert len(pb.losses) == 3 and len(pb.rx) == 2 and len(pb.ry) == 1
    assert pb.rowptr[0] == 0 and pb.rowptr[-1] == 6 and pb.colptr[-1] == 8


def test_bool_label_coercion():
    A = np.array([[1, 0, -1], [True, False, True]], dtype=object)
    g = L.GLRM(A, L.LogisticLoss(), L.ZeroReg(), L.ZeroReg(), 1)
    np.testing.assert_array_equal(g.problem_arrays().rowvals, [1, 0, 0, 1, 0, 1])


def test_fit_inplace_warm_start_and_shared_history(capsys):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((40, 3)) @ rng.standard_normal((3, 30))
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), 3, rng=rng)
    eng = O.oracle_api()
    X, Y, ch = L.fit_b(g, L.ProxGradParams(max_iter=20), engine=eng)
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "Fitting GLRM" and out[1].startswith("Iteration 10: objective value = ")
    assert X is g.X and Y is g.Y  # mutated in place, returned by reference (proxgrad.jl:219)
    assert len(ch.objective) == len(ch.times) == 21 and ch.times[0] == 0 and np.all(np.diff(ch.times) >= 0)
    # a second call continues from the current estimate and appends to the same history (cross_validate.jl:174)
    n0, last = len(ch.objective), ch.objective[-1]
    L.fit_b(g, L.ProxGradParams(max_iter=5), ch=ch, verbose=False, engine=eng)
    assert len(ch.objective) == n0 + 6
    assert ch.objective[-1] < last
    # non-mutating fit returns X' and restores the estimate (src/fit.jl:24-31)
    Xb, Yb = g.X.copy(), g.Y.copy()
    Xt, Y2, _ = L.fit(g, L.ProxGradParams(max_iter=3), verbose=False, engine=eng)
    assert Xt.shape == (40, 3) and np.array_equal(g.X, Xb) and np.array_equal(g.Y, Yb)
    assert not np.array_equal(Xt.T, Xb)


def test_engine_error_codes():
    from lowrankmodels.jl_amd import _capi
    rng = np.random.default_rng(2)
    A = rng.standard_normal((6, 5))
    g = L.GLRM(A, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), 2, rng=rng)
    eng = O.oracle_api()
    g.Y[...] = 0
    with pytest.raises(ValueError, match="all zeros"):
        L.fit_b(g, engine=eng, verbose=False)
    pa = g.problem_arrays()
    pa.rowvals = pa.rowvals.copy()
    pa.rowvals[3] = np.nan
    with pytest.raises(_capi.GLRMError) as ei:
        eng.create(pa)
    assert ei.value.code == _capi.ERR_NONFINITE and "is NaN" in ei.value.message
    pa = g.problem_arrays()
    pa.losses = pa.losses.copy()
    pa.losses["kind"] = 42
    with pytest.raises(_capi.GLRMError) as ei:
        eng.create(pa)
    assert ei.value.code == _capi.ERR_UNSUPPORTED
    pa = g.problem_arrays()
    pa.colidx = pa.colidx.copy()
    pa.colidx[0] = 99
    with pytest.raises(_capi.GLRMError) as ei:
        eng.create(pa)
    assert ei.value.code == _capi.ERR_INVALID


def test_regularization_path_reuses_the_engine_handle():
    """regularization_path pattern (src/cross_validate.jl:228-231): scale_regularizer! then a warm-started fit.  The host
    keeps the engine handle (Omega / A stay resident) and only swaps the regularizer descriptors."""
    rng = np.random.default_rng(4)
    A = rng.standard_normal((50, 3)) @ rng.standard_normal((3, 35)) + 0.1 * rng.standard_normal((50, 35))
    I, J = np.nonzero(rng.random((50, 35)) < 0.6)
    X0, Y0 = rng.standard_normal((3, 50)), rng.standard_normal((3, 35))
    eng = O.oracle_api()
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(1.0), L.QuadReg(1.0), 3, obs=(I, J), X=X0, Y=Y0)
    p = L.ProxGradParams(max_iter=15)
    L.fit_b(g, p, verbose=False, engine=eng)
    h0 = g._handle_cache[1].value
    L.scale_regularizer_(g, 0.25)
    L.fit_b(g, p, verbose=False, engine=eng)
    assert g._handle_cache[1].value == h0  # same handle, descriptors replaced through glrm_*_set_regularizers
    # reference: the same two fits with fresh models
    r = L.GLRM(A, L.QuadLoss(), L.QuadReg(1.0), L.QuadReg(1.0), 3, obs=(I, J), X=X0, Y=Y0)
    L.fit_b(r, p, verbose=False, engine=eng)
    r2 = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.25), L.QuadReg(0.25), 3, obs=(I, J), X=r.X.copy(), Y=r.Y.copy())
    L.fit_b(r2, p, verbose=False, engine=eng)
    assert np.array_equal(g.X, r2.X) and np.array_equal(g.Y, r2.Y)
    g.close()


def test_descriptor_cache_tracks_every_way_of_changing_a_model():
    """The packed descriptors of a model are cached (a million Python objects are not re-read per fit! call); any construction or
    modification of a loss / regularizer and any list mutation invalidates the cache."""
    rng = np.random.default_rng(0)
    g = L.GLRM(rng.standard_normal((30, 8)), L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.2), 2)
    k0 = g._descriptor_key()
    assert g._descriptor_key() is k0                       # cached object
    g.rx[3].scale = 0.7                                    # attribute assignment on one regularizer
    k1 = g._descriptor_key()
    assert k1 != k0 and k1[0][1] == 30                     # one descriptor per row now (a uniform list packs to one)
    g.ry[2] = L.OneReg(0.3)                                # list element replaced
    k2 = g._descriptor_key()
    assert k2[1] != k1[1]
    g.losses[1].mul_(3.0)                                  # loss scale: forces a new engine handle
    assert g._descriptor_key()[0] != k2[0]
    L.scale_regularizer_(g, 2.0)
    k3 = g._descriptor_key()
    g.rx = [L.ZeroReg()] * 30                              # a plain list: no tracking, the key is recomputed every time
    assert g._descriptor_key()[1] != k3[1] and g._descriptor_key() is not g._descriptor_key()


def test_simple_glrm_builders():
    """src/simple_glrms.jl; test/runtests.jl:19-25 runs kmeans with inner_iter=10 on two separated clusters."""
    rng = np.random.default_rng(0)
    A = np.vstack([rng.standard_normal((100, 10)) + 5, rng.standard_normal((50, 10)) - 5])
    g = E.kmeans(A, 2, rng=rng)
    X, Y, ch = L.fit_b(g, L.ProxGradParams(max_iter=30, inner_iter=10), verbose=False, engine=O.oracle_api())
    lab = np.argmax(X, axis=0)
    assert len(set(lab[:100])) == 1 and len(set(lab[100:])) == 1 and lab[0] != lab[-1]      # 100 / 50 split recovered
    for build, rx, ry in ((E.pca, L.ZeroReg, L.ZeroReg), (E.nnmf, L.NonNegConstraint, L.NonNegConstraint)):
        m = build(A, 3, rng=rng)
        assert isinstance(m.rx[0], rx) and isinstance(m.ry[0], ry) and isinstance(m.losses[0], L.QuadLoss)
    q, r = E.qpca(A, 3, scale=0.5, rng=rng), E.rpca(A, 3, scale=2.0, rng=rng)
    assert q.rx[0].scale == 0.5 and isinstance(r.losses[0], L.HuberLoss) and r.ry[0].scale == 2.0


def test_add_offset_after_a_first_fit_reuses_the_handle_on_the_general_path():
    """fit!, then add_offset! (src/modify_glrm.jl:20-25), then fit! again on the SAME model: the cached engine handle only gets new
    regularizer descriptors (set_regularizers) and must move to the general path that knows lastentry1 / lastentry_unpenalized --
    on the oracle engine too (it is the checker of that flow for the HIP engine)."""
    rng = np.random.default_rng(77)
    m, n, k = 60, 25, 4
    A = rng.standard_normal((m, 3)) @ rng.standard_normal((3, n)) + 2.0
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    p = L.ProxGradParams(max_iter=8)
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), k, X=X0.copy(), Y=Y0.copy())
    L.fit_b(g, p, verbose=False, engine=O.oracle_api())
    Xw, Yw = g.X.copy(), g.Y.copy()
    L.add_offset_(g)
    L.fit_b(g, p, verbose=False, engine=O.oracle_api())      # cached handle + set_regularizers
    assert np.all(g.X[-1] == 1.0)
    f = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), k, X=Xw.copy(), Y=Yw.copy(), offset=True)
    L.fit_b(f, p, verbose=False, engine=O.oracle_api())      # fresh handle created with the wrappers
    assert np.array_equal(g.X, f.X) and np.array_equal(g.Y, f.Y)


def test_copy_estimate_builds_its_own_train_test_split():
    """copy_estimate shares the problem data and copies X, Y (src/conveniencemethods.jl:16-20): a train / test split taken from the
    copy carries the COPY's factors even after the original has cached a split of its own."""
    rng = np.random.default_rng(78)
    A = rng.standard_normal((30, 12))
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), 3, X=rng.standard_normal((3, 30)), Y=rng.standard_normal((3, 12)))
    L.get_train_and_test(g, 0.2, rng=np.random.default_rng(1), engine=O.oracle_api(), fused=False)  # caches a split on g
    c = L.copy_estimate(g)
    c.X[...] = 7.0
    train, test = L.get_train_and_test(c, 0.2, rng=np.random.default_rng(1), engine=O.oracle_api(), fused=False)
    assert np.all(train.X == 7.0) and np.all(test.X == 7.0) and not np.any(g.X == 7.0)
