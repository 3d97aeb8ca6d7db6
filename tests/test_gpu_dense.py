"""-m gpu: the fully observed QuadLoss hand-over (`glrm_problem.dense_A`, fp64 MFMA half-steps) against the
oracle run on the equivalent explicit observation lists."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi

pytestmark = pytest.mark.gpu
TOL = 1e-5


def hip():
    return _capi.hip_api()


def dense_case(rng, m, n, k, rx, ry, scale=1.0, noise=0.1):
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) / np.sqrt(k) + noise * rng.standard_normal((m, n))
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    g = L.GLRM(A, L.QuadLoss(scale), rx, ry, k, X=X0, Y=Y0)
    assert g.dense_eligible()
    return g, np.asfortranarray(X0), np.asfortranarray(Y0)


def compare_dense(g, X0, Y0, params, quad_gram=0):
    """quad_gram = 1: glrm_options.quad_gram -- the line-search trials come from the quadratic form J(x) + g.s + scale s'(YY')s instead
    of a pass over A; same tolerances against the oracle (which evaluates every trial like the reference)."""
    O.set_threads(4)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), g.problem_arrays(), X0, Y0, params)
    o_g, X_g, Y_g, st_g = cases.run_engine(hip(), g.problem_arrays(dense=True), X0, Y0, params, quad_gram=quad_gram)
    assert st_g["tiled"] & 4, "the dense MFMA path was not taken"
    assert len(o_g) == len(o_c)
    e = (cases.rel_err(o_g, o_c), cases.fro_err(X_g, X_c), cases.fro_err(Y_g, Y_c))
    assert max(e) < TOL, e
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert abs(st_g[key] - st_c[key]) <= max(5, 0.03 * st_c[key]), (key, st_g[key], st_c[key])
    assert st_g["nnz_rows"] == st_g["nnz_cols"] == g.m * g.n
    return e


@pytest.mark.parametrize("m,n,k", [(150, 90, 9), (64, 64, 16), (333, 257, 32), (100, 100, 33), (200, 130, 64), (17, 1000, 20),
                                   (1000, 17, 12)])
@pytest.mark.parametrize("quad_gram", [0, 1])
def test_dense_path_matches_oracle(m, n, k, quad_gram):
    rng = np.random.default_rng(1000 + m + n + k)
    g, X0, Y0 = dense_case(rng, m, n, k, L.QuadReg(0.1), L.QuadReg(0.1))
    compare_dense(g, X0, Y0, L.ProxGradParams(max_iter=15), quad_gram)


@pytest.mark.parametrize("name", ["zero", "nonneg", "one", "scaled"])
@pytest.mark.parametrize("quad_gram", [0, 1])
def test_dense_path_regularizers_and_scale(name, quad_gram):
    rng = np.random.default_rng(5)
    rx, ry, scale = {"zero": (L.ZeroReg(), L.ZeroReg(), 1.0), "nonneg": (L.NonNegConstraint(), L.NonNegConstraint(), 1.0),
                     "one": (L.OneReg(0.2), L.QuadReg(0.3), 1.0), "scaled": (L.QuadReg(0.1), L.ZeroReg(), 2.5)}[name]
    g, X0, Y0 = dense_case(rng, 180, 140, 24, rx, ry, scale)
    compare_dense(g, X0, Y0, L.ProxGradParams(max_iter=20), quad_gram)


@pytest.mark.parametrize("quad_gram", [0, 1])
def test_dense_per_row_regularizers_and_inner_iterations(quad_gram):
    rng = np.random.default_rng(6)
    kinds = [L.QuadReg(0.3), L.OneReg(0.2), L.NonNegConstraint(), L.ZeroReg()]
    m, n, k = 120, 70, 16
    g, X0, Y0 = dense_case(rng, m, n, k, [kinds[i % 4] for i in range(m)], L.QuadReg(0.05))
    compare_dense(g, X0, Y0, L.ProxGradParams(max_iter=8, inner_iter=3), quad_gram)


def test_host_level_fit_takes_the_dense_path_and_agrees_with_lists():
    rng = np.random.default_rng(7)
    m, n, k = 300, 200, 16
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n))
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    mk = lambda: L.GLRM(A, L.QuadLoss(), L.QuadReg(.1), L.QuadReg(.1), k, X=X0, Y=Y0)
    gd, gl, gc = mk(), mk(), mk()
    _, _, chd = L.fit_b(gd, L.HipProxGradParams(max_iter=30), verbose=False)                 # dense hand-over (default)
    _, _, chl = L.fit_b(gl, L.HipProxGradParams(max_iter=30, dense=False), verbose=False)    # explicit lists
    _, _, chc = L.fit_b(gc, L.ProxGradParams(max_iter=30), verbose=False, engine=O.oracle_api())
    assert gd._handle_cache[0].kernel_stats(gd._handle_cache[1])["tiled"] & 4
    assert not gl._handle_cache[0].kernel_stats(gl._handle_cache[1])["tiled"] & 4
    for ch, g in ((chd, gd), (chl, gl)):
        assert len(ch.objective) == len(chc.objective)
        assert cases.rel_err(ch.objective, chc.objective) < TOL and cases.fro_err(g.X, gc.X) < TOL and cases.fro_err(g.Y, gc.Y) < TOL
    assert L.objective(gd) == pytest.approx(L.objective(gc, engine=O.oracle_api()), rel=1e-10)
    gd.close(); gl.close()


def test_quad_gram_follows_the_pass_over_A_to_convergence():
    """glrm_options.quad_gram against the default on a run to the reference's own stop rule (host-level API): same number of
    iterations, objective and factors to 1e-9 -- and against the oracle to the usual tolerance."""
    rng = np.random.default_rng(17)
    m, n, k = 400, 300, 32
    A = rng.standard_normal((m, 6)) @ rng.standard_normal((6, n)) + 0.05 * rng.standard_normal((m, n))
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    mk = lambda: L.GLRM(A, L.QuadLoss(), L.QuadReg(.05), L.NonNegConstraint(), k, X=X0, Y=Y0)
    ga, gg, gc = mk(), mk(), mk()
    _, _, cha = L.fit_b(ga, L.HipProxGradParams(max_iter=200), verbose=False)
    _, _, chg = L.fit_b(gg, L.HipProxGradParams(max_iter=200, quad_gram=True), verbose=False)
    _, _, chc = L.fit_b(gc, L.ProxGradParams(max_iter=200), verbose=False, engine=O.oracle_api())
    assert gg._handle_cache[0].kernel_stats(gg._handle_cache[1])["tiled"] & 4
    assert len(chg.objective) == len(cha.objective) == len(chc.objective) and len(chg.objective) > 20
    assert cases.rel_err(chg.objective, cha.objective) < 1e-9 and cases.fro_err(gg.X, ga.X) < 1e-9 and cases.fro_err(gg.Y, ga.Y) < 1e-9
    assert cases.rel_err(chg.objective, chc.objective) < TOL and cases.fro_err(gg.X, gc.X) < TOL
    ga.close(); gg.close()


@pytest.mark.parametrize("quad_gram", [0, 1])
def test_dense_two_shards_equal_one_shard(quad_gram):
    import torch
    rng = np.random.default_rng(8)
    m, n, k = 260, 150, 32
    g, X0, Y0 = dense_case(rng, m, n, k, L.QuadReg(0.1), L.NonNegConstraint())
    api, params = hip(), L.ProxGradParams(max_iter=6)
    o1, X1, Y1, _ = cases.run_engine(api, g.problem_arrays(dense=True), X0, Y0, params, quad_gram=quad_gram)
    stream = torch.cuda.current_stream().cuda_stream
    dev = torch.device("cuda", 0)
    hs = [api.create(g.problem_arrays(rows=(0, 100), cols=(0, 70), dense=True), stream=stream, quad_gram=quad_gram),
          api.create(g.problem_arrays(rows=(100, m), cols=(70, n), dense=True), stream=stream, quad_gram=quad_gram)]
    ld = api.factor_ld(hs[0])
    dX, dY = torch.zeros(m * ld, dtype=torch.float64, device=dev), torch.zeros(n * ld, dtype=torch.float64, device=dev)
    dC, dR = torch.zeros(n, dtype=torch.float64, device=dev), torch.zeros(m, dtype=torch.float64, device=dev)
    for h in hs:
        api.bind_buffers(h, dX.data_ptr(), dY.data_ptr(), dC.data_ptr(), dR.data_ptr())
    api.set_factors(hs[0], X0, Y0)
    for h in hs:
        api.reset_stepsizes(h, params.stepsize)
    objs = []
    for _ in range(params.max_iter):
        for h in hs:
            api.step_x(h, params.min_stepsize)
        for h in hs:
            api.step_y(h, params.min_stepsize)
        objs.append(api.sum(hs[0], dC.data_ptr(), n))
    X2, Y2 = np.zeros_like(X0), np.zeros_like(Y0)
    api.get_factors(hs[0], X2, Y2)
    for h in hs:
        api.destroy(h)
    assert np.array_equal(X1, X2) and np.array_equal(Y1, Y2) and np.array_equal(o1[1:], np.array(objs))


def test_dense_argument_errors():
    rng = np.random.default_rng(9)
    g, X0, Y0 = dense_case(rng, 40, 30, 16, L.ZeroReg(), L.ZeroReg())
    api = hip()
    pa = g.problem_arrays(dense=True)
    lists = g.problem_arrays()
    pa.rowptr, pa.colidx, pa.rowvals = lists.rowptr, lists.colidx, lists.rowvals
    with pytest.raises(_capi.GLRMError) as ei:
        api.create(pa)
    assert ei.value.code == _capi.ERR_INVALID
    pa = g.problem_arrays(dense=True)
    pa.losses = np.array([(1, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)  # L1Loss: not a dense-path loss
    with pytest.raises(_capi.GLRMError) as ei:
        api.create(pa)
    assert ei.value.code == _capi.ERR_UNSUPPORTED
    pa = g.problem_arrays(dense=True)
    bad = pa.dense_A.copy(); bad[3, 4] = np.nan
    pa.dense_A = bad
    with pytest.raises(_capi.GLRMError) as ei:
        api.create(pa)
    assert ei.value.code == _capi.ERR_NONFINITE and "(3, 4)" in ei.value.message
    # glrm_options.reserved must be 0 (the field behind quad_gram)
    import ctypes
    o = _capi.COptions(-1, 0, 0, 0, None, 0, 0, 1, 7)
    h = ctypes.c_void_p()
    pa = g.problem_arrays(dense=True)
    cp = api._cproblem(pa)
    rc = api._f["create"](ctypes.byref(h), ctypes.byref(cp), ctypes.byref(o))
    assert rc == _capi.ERR_INVALID and not h.value


def test_quad_gram_is_ignored_outside_the_dense_path():
    """glrm_options.quad_gram on a handle with observation lists changes nothing (the Gram matrix of a sparse row is its own)."""
    kwargs, params = cases.build_golden_case("c1")
    pa = L.GLRM(**kwargs).problem_arrays()
    o0, X0_, Y0_, _ = cases.run_engine(hip(), pa, kwargs["X"], kwargs["Y"], params)
    o1, X1_, Y1_, st = cases.run_engine(hip(), pa, kwargs["X"], kwargs["Y"], params, quad_gram=1)
    assert not st["tiled"] & 4 and np.array_equal(o0, o1) and np.array_equal(X0_, X1_) and np.array_equal(Y0_, Y1_)
