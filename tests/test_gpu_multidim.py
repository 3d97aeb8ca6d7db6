"""-m gpu: the general sweeps of the HIP engine (multi-dimensional losses, block regularizers, offsets; csrc/glrm_multi.hpp)
against the CPU oracle through the C ABI.  Tolerance 1e-5 relative on trajectories and factors (north star); the
per-entry arithmetic is the same expression tree, so the observed agreement is ~1e-12."""
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


def compare(pa, X0, Y0, params, tol=TOL):
    O.set_threads(4)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, params)
    o_g, X_g, Y_g, st_g = cases.run_engine(hip(), pa, X0, Y0, params)
    assert st_g["tiled"] == 8  # routed to the general sweeps
    assert len(o_g) == len(o_c)
    e = (cases.rel_err(o_g, o_c), cases.fro_err(X_g, X_c), cases.fro_err(Y_g, Y_c))
    assert max(e) < tol, e
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert abs(st_g[key] - st_c[key]) <= max(5, 0.03 * st_c[key]), (key, st_g[key], st_c[key])
    return e


@pytest.mark.parametrize("name", list(cases.MULTIDIM_CASES))
def test_multidim_cases_match_oracle(name):
    kwargs, p = cases.build_multidim_case(name)
    compare(L.GLRM(**kwargs).problem_arrays(), kwargs["X"], kwargs["Y"], p)


def random_categorical(rng, m, n, k, dmax, density=0.6, ordinal=False):
    losses, ry = [], []
    for f in range(n):
        d = int(rng.integers(2, dmax + 1))
        kind = f % 6
        if kind == 0:
            losses.append(L.MultinomialLoss(d)); ry.append(L.QuadReg(0.1))
        elif kind == 1:
            losses.append(L.OvALoss(d, bin_loss=L.HingeLoss() if f % 4 == 1 else L.LogisticLoss())); ry.append(L.OneReg(0.05))
        elif kind == 2:
            losses.append(L.BvSLoss(d + 1)); ry.append(L.OrdinalReg(L.QuadReg(0.1)) if ordinal else L.QuadReg(0.1))
        elif kind == 3:
            losses.append(L.MultinomialOrdinalLoss(d + 1)); ry.append(L.MNLOrdinalReg(L.QuadReg(0.05)) if ordinal else L.QuadReg(0.05))
        elif kind == 4:
            losses.append(L.OrdisticLoss(d)); ry.append(L.lastentry_unpenalized(L.QuadReg(0.1)) if ordinal else L.QuadReg(0.2))
        else:
            losses.append(L.HuberLoss()); ry.append(L.lastentry_unpenalized(L.QuadReg(0.1)) if ordinal else L.QuadReg(0.1))
    A = np.zeros((m, n))
    Z = rng.standard_normal((m, 3))
    for f, lo in enumerate(losses):
        z = Z @ rng.standard_normal(3)
        A[:, f] = cases._levels(z, 1, lo.max) if hasattr(lo, "max") else z
    I, J = np.nonzero(rng.random((m, n)) < density)
    D = L.embedding_dim(losses)
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, D))
    rx = L.lastentry1(L.QuadReg(0.1)) if ordinal else L.QuadReg(0.1)
    return L.GLRM(A, losses, rx, ry, k, obs=(I, J), X=X0, Y=Y0), X0, Y0


@pytest.mark.parametrize("k,dmax,ordinal", [(1, 3, False), (2, 4, True), (5, 6, True), (8, 8, False), (13, 10, True), (16, 5, False),
                                           (33, 7, True), (64, 32, False), (60, 31, True)])
def test_random_categorical_models(k, dmax, ordinal):
    """Ranks across the factor leading dimensions (8 / 16 / 32 / 64) and embedding dimensions up to the ABI limit of 32."""
    rng = np.random.default_rng(1000 + 7 * k + dmax)
    g, X0, Y0 = random_categorical(rng, 150, 24, k, dmax, ordinal=ordinal)
    compare(g.problem_arrays(), X0, Y0, L.ProxGradParams(max_iter=8))


def test_long_columns_use_all_waves():
    """A tall model (3000 observations per column): the 8 waves of a column workgroup each take a slice of the list."""
    rng = np.random.default_rng(77)
    g, X0, Y0 = random_categorical(rng, 5000, 12, 6, 5, density=0.6, ordinal=True)
    compare(g.problem_arrays(), X0, Y0, L.ProxGradParams(max_iter=6))


def test_split_column_sweeps_on_a_tall_model():
    """Columns longer than one chunk (8192 observations) are swept by several workgroups per column (pass + decide rounds)."""
    rng = np.random.default_rng(78)
    g, X0, Y0 = random_categorical(rng, 30000, 6, 5, 4, density=0.7, ordinal=True)
    compare(g.problem_arrays(), X0, Y0, L.ProxGradParams(max_iter=5))


@pytest.mark.parametrize("name", ["categorical_mix", "ordinal_offsets", "loss_test"])
def test_split_and_one_kernel_column_sweeps_agree(name, monkeypatch):
    """GLRM_HIP_MULTI_CHUNK=8 forces the split path on the small cases: same trajectory as the one-kernel sweep (only the grouping
    of the partial sums differs), and two column shards reproduce the single handle bit for bit on the split path."""
    kwargs, p = cases.build_multidim_case(name)
    g = L.GLRM(**kwargs)
    api = hip()
    ref = cases.run_engine(api, g.problem_arrays(), kwargs["X"], kwargs["Y"], p)
    monkeypatch.setenv("GLRM_HIP_MULTI_CHUNK", "8")
    spl = cases.run_engine(api, g.problem_arrays(), kwargs["X"], kwargs["Y"], p)
    assert len(spl[0]) == len(ref[0])
    assert cases.rel_err(spl[0], ref[0]) < 1e-9 and cases.fro_err(spl[1], ref[1]) < 1e-9 and cases.fro_err(spl[2], ref[2]) < 1e-9
    # sharded, split path: one Y half-step on two column blocks == on the whole
    X, Y = np.array(kwargs["X"], order="F"), np.array(kwargs["Y"], order="F")
    ys = g.problem_arrays().ystart
    outs = []
    for bounds in ([0, g.n], [0, g.n // 3, g.n]):
        Yout = Y.copy()
        for r in range(len(bounds) - 1):
            h = api.create(g.problem_arrays(cols=(bounds[r], bounds[r + 1])))
            try:
                api.set_factors(h, X, Y)
                api.reset_stepsizes(h, 1.0)
                api.step_y(h, 0.01)
                Xr, Yr = np.zeros_like(X), np.zeros_like(Y)
                api.get_factors(h, Xr, Yr)
                Yout[:, ys[bounds[r]]:ys[bounds[r + 1]]] = Yr[:, ys[bounds[r]]:ys[bounds[r + 1]]]
            finally:
                api.destroy(h)
        outs.append(Yout)
    assert np.array_equal(outs[0], outs[1])


def test_rank_above_64_is_rejected_for_multidim():
    rng = np.random.default_rng(5)
    g, X0, Y0 = random_categorical(rng, 40, 6, 70, 3)
    with pytest.raises(_capi.GLRMError) as ei:
        hip().create(g.problem_arrays())
    assert ei.value.code == _capi.ERR_UNSUPPORTED


def test_bad_levels_are_rejected():
    kwargs, _ = cases.build_multidim_case("mnl")
    A = np.array(kwargs["A"]); A[3, 2] = 0.0
    kwargs["A"] = A
    with pytest.raises(_capi.GLRMError) as ei:
        hip().create(L.GLRM(**kwargs).problem_arrays())
    assert ei.value.code == _capi.ERR_NONFINITE


@pytest.mark.parametrize("name", ["categorical_mix", "ordinal_offsets", "loss_test"])
def test_step_level_shards_reproduce_whole_fit(name):
    """Two row/column shards driven through the step-level API give the single-handle result bit for bit; the Y
    exchange moves the vector span [ystart[cb], ystart[ce]) of each shard."""
    kwargs, p = cases.build_multidim_case(name)
    p = L.ProxGradParams(max_iter=6)
    g = L.GLRM(**kwargs)
    api = hip()
    obj_ref, X_ref, Y_ref, _ = cases.run_engine(api, g.problem_arrays(), kwargs["X"], kwargs["Y"], p)
    rb, cb = [0, g.m // 2, g.m], [0, g.n // 3, g.n]
    hs = [api.create(g.problem_arrays(rows=(rb[r], rb[r + 1]), cols=(cb[r], cb[r + 1]))) for r in range(2)]
    try:
        ys = g.problem_arrays().ystart
        X, Y = np.array(kwargs["X"], order="F"), np.array(kwargs["Y"], order="F")
        for h in hs:
            api.set_factors(h, X, Y)
            api.reset_stepsizes(h, p.stepsize)
        for it in range(p.max_iter):
            for h in hs:
                api.step_x(h, p.min_stepsize)
            parts = []
            for h in hs:
                Xr, Yr = np.zeros_like(X), np.zeros_like(Y)
                api.get_factors(h, Xr, Yr); parts.append(Xr)
            for r in range(2):
                X[:, rb[r]:rb[r + 1]] = parts[r][:, rb[r]:rb[r + 1]]
            for h in hs:
                api.set_factors(h, X, Y)
                api.step_y(h, p.min_stepsize)
            parts = []
            for h in hs:
                Xr, Yr = np.zeros_like(X), np.zeros_like(Y)
                api.get_factors(h, Xr, Yr); parts.append(Yr)
            for r in range(2):
                Y[:, ys[cb[r]]:ys[cb[r + 1]]] = parts[r][:, ys[cb[r]]:ys[cb[r + 1]]]
            for h in hs:
                api.set_factors(h, X, Y)
        assert np.array_equal(X, X_ref) and np.array_equal(Y, Y_ref)
    finally:
        for h in hs:
            api.destroy(h)


def test_step_x_range_chunks_equal_full_sweep():
    kwargs, p = cases.build_multidim_case("loss_test")
    g = L.GLRM(**kwargs)
    api = hip()
    outs = []
    for chunks in (1, 4):
        h = api.create(g.problem_arrays())
        try:
            X, Y = np.array(kwargs["X"], order="F"), np.array(kwargs["Y"], order="F")
            api.set_factors(h, X, Y)
            api.reset_stepsizes(h, 1.0)
            c = g.m // chunks
            for j in range(chunks):
                api.step_x_range(h, j * c, g.m if j == chunks - 1 else (j + 1) * c, 0.01)
            api.get_factors(h, X, Y)
            outs.append(X.copy())
        finally:
            api.destroy(h)
    assert np.array_equal(outs[0], outs[1])


def test_objective_and_penalties_match_oracle():
    for name in ("ordinal_offsets", "categorical_mix"):
        kwargs, _ = cases.build_multidim_case(name)
        g = L.GLRM(**kwargs)
        X, Y = np.array(kwargs["X"], order="F"), np.array(kwargs["Y"], order="F")
        X[-1, :] = 1.0  # make lastentry1 finite
        vals = []
        for api in (O.oracle_api(), hip()):
            h = api.create(g.problem_arrays())
            try:
                vals.append((api.objective(h, X, Y, True), api.objective(h, X, Y, False)))
            finally:
                api.destroy(h)
        assert np.isfinite(vals[0][0])
        assert vals[1][0] == pytest.approx(vals[0][0], rel=1e-12) and vals[1][1] == pytest.approx(vals[0][1], rel=1e-12)


def test_sparse_solver_on_multidim_model():
    kwargs, _ = cases.build_multidim_case("categorical_mix")
    g = L.GLRM(**kwargs)
    p = L.SparseProxGradParams(max_iter=25)
    res = []
    for api in (O.oracle_api(), hip()):
        h = api.create(g.problem_arrays())
        try:
            X, Y = np.array(kwargs["X"], order="F"), np.array(kwargs["Y"], order="F")
            obj, _ = api.fit_sparse(h, p, X, Y)
            res.append((obj, X, Y))
        finally:
            api.destroy(h)
    assert len(res[0][0]) == len(res[1][0])
    assert cases.rel_err(res[1][0], res[0][0]) < TOL
    assert cases.fro_err(res[1][1], res[0][1]) < TOL and cases.fro_err(res[1][2], res[0][2]) < TOL


def test_set_regularizers_moves_a_scalar_handle_to_the_general_sweeps():
    """add_offset! on a live model: the handle was created on the fast paths, the new descriptors carry wrap flags."""
    rng = np.random.default_rng(31)
    m, n, k = 200, 40, 4
    A = rng.standard_normal((m, 3)) @ rng.standard_normal((3, n)) + 2.0
    I, J = np.nonzero(rng.random((m, n)) < 0.5)
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), k, obs=(I, J), X=X0, Y=Y0)
    p = L.ProxGradParams(max_iter=10)
    api = hip()
    h = api.create(g.problem_arrays())
    try:
        assert api.kernel_stats(h)["tiled"] != 8
        L.add_offset_(g)
        from lowrankmodels.jl_amd.regularizers import pack_regs
        api.set_regularizers(h, pack_regs(g.rx), pack_regs(g.ry))
        assert api.kernel_stats(h)["tiled"] == 8
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        obj, _ = api.fit(h, p, X, Y)
    finally:
        api.destroy(h)
    o_c, X_c, Y_c, _ = cases.run_engine(O.oracle_api(), g.problem_arrays(), X0, Y0, p)
    assert cases.rel_err(obj, o_c) < TOL and cases.fro_err(X, X_c) < TOL and cases.fro_err(Y, Y_c) < TOL
    assert np.all(X[-1] == 1.0)


def test_python_fit_with_offset_and_categoricals():
    """The host-side mirror: GLRM(..., offset=True) + fit! on the HIP engine, warm start continues."""
    kwargs, _ = cases.build_multidim_case("categorical_mix")
    g = L.GLRM(offset=True, **kwargs)
    X, Y, ch = L.fit_b(g, L.HipProxGradParams(max_iter=15), verbose=False)
    assert Y.shape == (3, L.embedding_dim(g.losses)) and np.all(X[-1] == 1.0)
    assert ch.objective[-1] < ch.objective[1]
    start = L.objective(g)
    X2, Y2, ch2 = L.fit_b(g, L.HipProxGradParams(max_iter=5), verbose=False)
    assert ch2.objective[0] == pytest.approx(start, rel=1e-12)
    assert ch2.objective[-1] <= ch2.objective[0]


@pytest.mark.parametrize("model", ["mnl_only", "mnl_and_scalars", "ordinal"])
@pytest.mark.parametrize("tall", [False, True])
def test_kind_specialised_kernels_give_the_bits_of_the_general_ones(model, tall, monkeypatch):
    """Models whose columns are all MultinomialLoss, or MultinomialLoss + scalar losses (the categorical / real / boolean columns of a data
    frame), run kernels compiled without the other kinds' code (glrm_multi.hpp: MULTI_KM_*; 6 100 instead of 11 000 instructions, no
    scratch).  Same formulas in the same order: bit-identical to the all-kinds kernels (GLRM_HIP_MULTI_KINDS=0), and within 1e-5 of the
    oracle.  ordinal: BvSLoss + MultinomialOrdinalLoss columns under OrdinalReg / MNLOrdinalReg with lastentry1 on X (the reference's ordinal
    data frame, test/prob_tests/BvSLoss.jl, MultinomialOrdinalLoss.jl).  tall: columns long enough for the split column passes
    (multi_colpass_kernel)."""
    rng = np.random.default_rng(77 if tall else 78)
    m, n, k = (30000, 6, 6) if tall else (400, 24, 7)
    losses = []
    for f in range(n):
        if model == "ordinal":
            losses.append(L.MultinomialOrdinalLoss(int(rng.integers(3, 8))) if f % 2 else L.BvSLoss(int(rng.integers(3, 8))))
        elif model == "mnl_only" or f % 3 == 0:
            losses.append(L.MultinomialLoss(int(rng.integers(2, 7))))
        else:
            losses.append([L.QuadLoss(), L.LogisticLoss(), L.HuberLoss(), L.OrdinalHingeLoss(1, 5)][f % 4])
    Z = rng.standard_normal((m, 3))
    A = np.zeros((m, n))
    for f, lo in enumerate(losses):
        z = Z @ rng.standard_normal(3)
        if hasattr(lo, "max") and not isinstance(lo, L.OrdinalHingeLoss):
            A[:, f] = np.clip(np.round((1 + lo.max) / 2 + z), 1, lo.max)
        elif isinstance(lo, L.OrdinalHingeLoss):
            A[:, f] = np.clip(np.round(3 + z), 1, 5)
        else:
            A[:, f] = (z > 0) if lo.classification else z
    I, J = np.nonzero(rng.random((m, n)) < 0.8)
    D = L.embedding_dim(losses)
    X0, Y0 = 0.5 * rng.standard_normal((k, m)), 0.5 * rng.standard_normal((k, D))
    rx, ry = L.QuadReg(0.1), L.QuadReg(0.1)
    if model == "ordinal":
        rx = L.lastentry1(L.QuadReg(0.1))
        ry = [L.MNLOrdinalReg(L.QuadReg(0.1)) if f % 2 else L.OrdinalReg(L.QuadReg(0.1)) for f in range(n)]
    g = L.GLRM(A, losses, rx, ry, k, obs=(I, J), X=X0, Y=Y0)
    pa = g.problem_arrays()
    p = L.ProxGradParams(max_iter=5)
    X0, Y0 = np.asfortranarray(X0), np.asfortranarray(Y0)
    a = cases.run_engine(hip(), pa, X0, Y0, p)
    monkeypatch.setenv("GLRM_HIP_MULTI_KINDS", "0")
    b = cases.run_engine(hip(), pa, X0, Y0, p)
    assert a[3]["tiled"] == b[3]["tiled"] == 8
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    monkeypatch.delenv("GLRM_HIP_MULTI_KINDS")
    compare(pa, X0, Y0, p)
