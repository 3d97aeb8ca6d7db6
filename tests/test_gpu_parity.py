"""-m gpu: the HIP engine (through the C ABI) against the CPU oracle on the same seeded inputs, against the
committed golden fixtures, and -- at full BASELINE size -- through size-independent properties.

Tolerance (north star): 1e-5 relative on the objective trajectory and on the factors (Frobenius-relative);
index / observation bookkeeping bit-exact.  In practice the fp64 engine agrees to ~1e-12 because the
per-entry arithmetic is the same expression tree and only the summation order differs.
"""
import ctypes as C
import os

import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi

pytestmark = pytest.mark.gpu
TOL = 1e-5
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hip():
    return _capi.hip_api()


def compare(pa, X0, Y0, params, tol=TOL, **create_kw):
    """HIP engine vs oracle.  Unless the caller pins a kernel family, BOTH sweep implementations are checked:
    tiled=1 the gather sweeps, tiled=2 the LDS-tiled sweeps (used wherever the index lists are sorted)."""
    O.set_threads(4)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, params)
    modes = [create_kw.pop("tiled")] if "tiled" in create_kw else ([1] if ("waves_row" in create_kw or "waves_col" in create_kw) else [1, 2])
    worst = (0.0, 0.0, 0.0)
    for mode in modes:
        o_g, X_g, Y_g, st_g = cases.run_engine(hip(), pa, X0, Y0, params, tiled=mode, **create_kw)
        assert len(o_g) == len(o_c), (mode, len(o_g), len(o_c))
        e_obj, e_x, e_y = cases.rel_err(o_g, o_c), cases.fro_err(X_g, X_c), cases.fro_err(Y_g, Y_c)
        assert e_obj < tol and e_x < tol and e_y < tol, (mode, e_obj, e_x, e_y)
        # accept / reject agreement.  The decision is a strict `<` between two nearly equal sums:
        # once a segment has converged its trials are rounding-level coin flips, so totals agree closely, not exactly.
        for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
            assert abs(st_g[key] - st_c[key]) <= max(5, 0.03 * st_c[key]), (mode, key, st_g[key], st_c[key])
        assert st_g["nnz_rows"] == st_c["nnz_rows"] and st_g["nnz_cols"] == st_c["nnz_cols"]
        worst = tuple(max(a, b) for a, b in zip(worst, (e_obj, e_x, e_y)))
    return worst


@pytest.mark.parametrize("name", list(cases.GOLDEN_CASES))
def test_golden_fixtures(name):
    pa, X0, Y0, params, z = cases.load_case(os.path.join(GOLDEN, name + ".npz"))
    for mode in (0, 1, 2):  # auto, gather sweeps, LDS-tiled sweeps
        obj, X, Y, st = cases.run_engine(hip(), pa, X0, Y0, params, tiled=mode)
        assert len(obj) == len(z["objective"])
        assert cases.rel_err(obj, z["objective"]) < TOL
        assert cases.fro_err(X, z["X"]) < TOL and cases.fro_err(Y, z["Y"]) < TOL
    compare(pa, X0, Y0, params)


def random_problem(rng, m, n, k, density, losses=None, rx=None, ry=None, dup=False):
    losses = L.QuadLoss() if losses is None else losses
    rx = L.QuadReg(0.1) if rx is None else rx
    ry = L.QuadReg(0.1) if ry is None else ry
    Z = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) / np.sqrt(k)
    A = Z + 0.1 * rng.standard_normal((m, n))
    if density >= 1.0:
        obs = None
    else:
        I, J = np.nonzero(rng.random((m, n)) < density)
        if dup:
            extra = rng.integers(0, len(I), len(I) // 10)
            I, J = np.concatenate([I, I[extra]]), np.concatenate([J, J[extra]])
            perm = rng.permutation(len(I))
            I, J = I[perm], J[perm]
        obs = (I, J)
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    g = L.GLRM(A, losses, rx, ry, k, obs=obs, X=X0, Y=Y0)
    return g.problem_arrays(), np.asfortranarray(X0), np.asfortranarray(Y0)


@pytest.mark.parametrize("k", [1, 2, 5, 8, 9, 16, 31, 32, 33, 64, 100, 128])
def test_every_rank_layout(k):
    """k <= 8 / 16 / 32 / 64 / 128 select the (G,R) lane layouts; non-multiples exercise the zero padding."""
    rng = np.random.default_rng(100 + k)
    pa, X0, Y0 = random_problem(rng, 150, 90, k, 0.4)
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=12))


@pytest.mark.parametrize("wr,wc", [(1, 1), (4, 4), (8, 8), (1, 8), (4, 1)])
def test_waves_per_segment_variants(wr, wc):
    """1 / 4 / 8 wavefronts per row or column (cross-wave LDS combine) give the same trajectory."""
    rng = np.random.default_rng(7)
    pa, X0, Y0 = random_problem(rng, 400, 60, 32, 0.5, rx=L.NonNegConstraint(), ry=L.NonNegConstraint())
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=15), waves_row=wr, waves_col=wc)


def test_auto_wave_selection_long_columns():
    rng = np.random.default_rng(8)
    pa, X0, Y0 = random_problem(rng, 6000, 12, 16, 0.6)  # ~3600 obs per column -> 4 waves per column
    h = hip().create(pa)
    st = hip().kernel_stats(h)
    hip().destroy(h)
    assert st["waves_col"] == 4 and st["waves_row"] == 1 and st["ld"] == 16
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=10))


def test_unsorted_lists_fall_back_to_gather_sweeps():
    """The LDS-tiled sweeps need the entries of tile t before those of tile t+1 (any order inside a tile); an Omega that is
    permuted across tiles silently uses the gather sweeps, one that fits a single tile does not care about its order."""
    rng = np.random.default_rng(81)
    pa, X0, Y0 = random_problem(rng, 2500, 2500, 8, 0.02, dup=True)  # k=8: 1920 vectors per tile -> two tiles per view
    h = hip().create(pa, tiled=2)
    st = hip().kernel_stats(h)
    hip().destroy(h)
    assert st["tiled"] == 0
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=5), tiled=2)
    # GLRM_HIP_TILE_SORT=2: the engine tile-sorts its private copy of such lists (what the auto choice does for large problems)
    os.environ["GLRM_HIP_TILE_SORT"] = "2"
    try:
        h = hip().create(pa, tiled=2)
        st = hip().kernel_stats(h)
        hip().destroy(h)
        assert st["tiled"] == 3
        compare(pa, X0, Y0, L.ProxGradParams(max_iter=8), tiled=2)
    finally:
        del os.environ["GLRM_HIP_TILE_SORT"]
    pa, X0, Y0 = random_problem(rng, 300, 80, 8, 0.4, dup=True)      # everything inside one tile: order is irrelevant
    h = hip().create(pa, tiled=2)
    st = hip().kernel_stats(h)
    hip().destroy(h)
    assert st["tiled"] == 3
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=8), tiled=2)
    pa, X0, Y0 = random_problem(rng, 300, 80, 8, 0.4)
    h = hip().create(pa, tiled=2)
    st = hip().kernel_stats(h)
    hip().destroy(h)
    assert st["tiled"] == 3


def test_tile_sort_in_batches_of_whole_segments():
    """hipCUB counts in int: views beyond 2e9 entries are tile-sorted batch by batch (glrm_tilesort.hip).  The batching is exercised
    here with a tiny batch limit (GLRM_HIP_TILE_SORT_BATCH) on shuffled lists with duplicates."""
    rng = np.random.default_rng(82)
    pa, X0, Y0 = random_problem(rng, 2500, 2500, 8, 0.02, dup=True)
    os.environ["GLRM_HIP_TILE_SORT"] = "2"
    os.environ["GLRM_HIP_TILE_SORT_BATCH"] = "3000"  # ~55 entries per segment: ~50 segments per batch
    try:
        h = hip().create(pa, tiled=2)
        st = hip().kernel_stats(h)
        hip().destroy(h)
        assert st["tiled"] == 3
        compare(pa, X0, Y0, L.ProxGradParams(max_iter=8), tiled=2)
    finally:
        del os.environ["GLRM_HIP_TILE_SORT"], os.environ["GLRM_HIP_TILE_SORT_BATCH"]


LOSS_CASES = {
    "l1": L.L1Loss(1.3), "huber": L.HuberLoss(0.9, crossover=0.6), "quantile": L.QuantileLoss(1.1, quantile=0.3),
    "periodic": L.PeriodicLoss(2.5, 0.8), "quad_scaled": L.QuadLoss(2.5),
}


@pytest.mark.parametrize("name", list(LOSS_CASES))
def test_real_valued_losses(name):
    rng = np.random.default_rng(31)
    pa, X0, Y0 = random_problem(rng, 120, 70, 6, 0.5, losses=LOSS_CASES[name], rx=L.QuadReg(0.05), ry=L.OneReg(0.05))
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=20))


def test_classification_count_and_ordinal_losses():
    rng = np.random.default_rng(32)
    m, n, k = 140, 60, 4
    Z = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) / 2
    A = np.zeros((m, n))
    losses = []
    for f in range(n):
        r = f % 4
        if r == 0:
            A[:, f] = rng.random(m) < 1 / (1 + np.exp(-Z[:, f])); losses.append(L.LogisticLoss(0.7))
        elif r == 1:
            A[:, f] = rng.random(m) < 0.5; losses.append(L.WeightedHingeLoss(1.2, case_weight_ratio=2.0))
        elif r == 2:
            A[:, f] = rng.poisson(np.exp(np.clip(Z[:, f], -2, 1.5))); losses.append(L.PoissonLoss())
        else:
            A[:, f] = np.clip(np.round(4 + 2 * Z[:, f]), 1, 7); losses.append(L.OrdinalHingeLoss(1, 7, 0.9))
    I, J = np.nonzero(rng.random((m, n)) < 0.6)
    X0, Y0 = 0.3 * rng.standard_normal((k, m)), 0.3 * rng.standard_normal((k, n))
    g = L.GLRM(A, losses, L.QuadReg(0.5), L.QuadReg(0.5), k, obs=(I, J), X=X0, Y=Y0)
    compare(g.problem_arrays(), np.asfortranarray(X0), np.asfortranarray(Y0), L.ProxGradParams(max_iter=25))
    # homogeneous non-quadratic loss takes the segment-uniform kernel variant for rows as well
    g2 = L.GLRM((A[:, ::4] > 0).astype(float), L.LogisticLoss(), L.QuadReg(0.1), L.QuadReg(0.1), k, X=X0, Y=Y0[:, ::4].copy())
    compare(g2.problem_arrays(), np.asfortranarray(X0), np.asfortranarray(Y0[:, ::4]), L.ProxGradParams(max_iter=15))


def test_per_row_and_per_column_regularizers():
    rng = np.random.default_rng(33)
    m, n, k = 90, 50, 5
    kinds = [L.QuadReg(0.3), L.OneReg(0.2), L.NonNegConstraint(), L.ZeroReg(), L.UnitOneSparseConstraint()]
    rx = [kinds[i % 5] for i in range(m)]
    ry = [kinds[(i * 3) % 4] for i in range(n)]
    pa, X0, Y0 = random_problem(rng, m, n, k, 0.5, rx=rx, ry=ry)
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=20))


def test_duplicates_empty_segments_and_single_entries():
    rng = np.random.default_rng(34)
    m, n, k = 80, 40, 7
    pa, X0, Y0 = random_problem(rng, m, n, k, 0.3, dup=True)
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=15))
    A = rng.standard_normal((m, n))
    feats = [[] for _ in range(m)]
    exs = [[] for _ in range(n)]
    feats[3] = [5]; feats[10] = [0, 0, 0, 39]; feats[79] = list(range(n))  # rows 0..2 etc. stay empty
    exs[5] = [3]; exs[0] = [10, 10, 10]; exs[39] = list(range(m)) + [10]
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.OneReg(0.1), k, observed_features=feats, observed_examples=exs, X=X0, Y=Y0)
    compare(g.problem_arrays(), np.asfortranarray(X0), np.asfortranarray(Y0), L.ProxGradParams(max_iter=15))


def test_moderate_sparse_shape_k32():
    """10^4 x 10^3, rank 32, 5 % observed, QuadReg (the C2 recipe at 1/1000 of the size)."""
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(10000, 1000, 32, 50)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(10000, 1000, 32, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=12))


def _c4_problem(m, n, q, k=64):
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=1)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(3, 0, 1.0)], dtype=_capi.REG_DTYPE)  # NonNegConstraint, src/regularizers.jl:101-114
    return _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg), X0, Y0


@pytest.mark.parametrize("start", ["nonneg", "randn"])
def test_c4_recipe(start):
    """BASELINE config 4 (the north-star target: rank 64, QuadLoss, NonNegConstraint on X and Y, 100 sorted observations per row,
    non-negative value model) at 20 000 x 2 000, both kernel families against the oracle.  `nonneg` is bench.py's start
    (|N(0,1)|/sqrt(k)); `randn` is the reference default (src/glrm.jl:31), whose objective starts at Inf and which collapses to
    X = 0 on this shape -- the engine has to follow the reference there too (Inf handling, then trials that all reject)."""
    pa, X0, Y0 = _c4_problem(20000, 2000, 100)
    if start == "nonneg":
        X0, Y0 = np.asfortranarray(np.abs(X0) / 8.0), np.asfortranarray(np.abs(Y0) / 8.0)
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=12))
    if start == "randn":
        o, X, Y, st = cases.run_engine(hip(), pa, X0, Y0, L.ProxGradParams(max_iter=12))
        assert np.isinf(o[0]) and np.count_nonzero(X) == 0  # the collapse is the reference's behaviour, not an engine artefact


def test_c4_recipe_sparse_rows_full_density():
    """The same recipe at the full problem's density (0.1 % observed: 100 observations per row over 100 000 columns, so the LDS tiles
    see ~0.3 observations per row and the auto choice must stay on the gather sweeps), 3 000 rows."""
    pa, X0, Y0 = _c4_problem(3000, 100000, 100)
    X0, Y0 = np.asfortranarray(np.abs(X0) / 8.0), np.asfortranarray(np.abs(Y0) / 8.0)
    h = hip().create(pa)
    st = hip().kernel_stats(h)
    hip().destroy(h)
    assert st["tiled"] == 0 and st["ld"] == 64
    compare(pa, X0, Y0, L.ProxGradParams(max_iter=6), tiled=0)


def test_mixed_losses_k32_several_tiles():
    """The C5 recipe at 1/2000 of the size: 2500 x 2000, rank 32, Quad / Logistic / OrdinalHinge columns, 100 observations per row.
    2000 columns = 4 LDS tiles, so batches of four observations straddle tile windows and loss kinds in the row sweep."""
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(2500, 2000, 32, 100, value_model=0, loss_mix=1)
    kinds = [L.QuadLoss().descriptor(), L.LogisticLoss().descriptor(), L.OrdinalHingeLoss(1, 5).descriptor()]
    losses = np.array([kinds[f % 3] for f in range(2000)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(2500, 2000, 32, rowptr, colidx, rowvals, colptr, rowidx, colvals, losses, reg, reg)
    compare(pa, 0.3 * X0, 0.3 * Y0, L.ProxGradParams(max_iter=10))


@pytest.mark.parametrize("k", [40, 64])
def test_mixed_losses_eight_lane_layout(k):
    """k = 33...64 puts eight lanes on an observation; the non-quadratic step then takes the batch of eight observations at once."""
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(1200, 900, k, 60, value_model=0, loss_mix=1)
    kinds = [L.QuadLoss().descriptor(), L.LogisticLoss().descriptor(), L.OrdinalHingeLoss(1, 5).descriptor()]
    losses = np.array([kinds[f % 3] for f in range(900)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(1200, 900, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, losses, reg, reg)
    compare(pa, 0.3 * X0, 0.3 * Y0, L.ProxGradParams(max_iter=10))
    # one non-quadratic loss for every column: the segment-uniform variant
    hub = np.array([L.HuberLoss(1.0, crossover=0.5).descriptor()], dtype=_capi.LOSS_DTYPE)
    pa2 = _capi.ProblemArrays(1200, 900, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, hub, reg, reg)
    compare(pa2, 0.3 * X0, 0.3 * Y0, L.ProxGradParams(max_iter=8))


def test_objective_entry_point():
    rng = np.random.default_rng(35)
    pa, X0, Y0 = random_problem(rng, 70, 30, 4, 0.5, rx=L.OneReg(0.3))
    for include in (True, False):
        vals = []
        for api in (hip(), O.oracle_api()):
            h = api.create(pa)
            vals.append(api.objective(h, X0, Y0, include))
            api.destroy(h)
        assert vals[0] == pytest.approx(vals[1], rel=1e-12)


def test_host_level_fit_and_handle_reuse():
    rng = np.random.default_rng(36)
    A = rng.standard_normal((60, 4)) @ rng.standard_normal((4, 45))
    X0, Y0 = rng.standard_normal((4, 60)), rng.standard_normal((4, 45))
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), 4, X=X0, Y=Y0)
    gc = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), 4, X=X0, Y=Y0)
    p = L.HipProxGradParams(max_iter=1)
    ch, chc = L.ConvergenceHistory("cv_by_iter"), L.ConvergenceHistory("cv_by_iter")
    for _ in range(5):  # cv_by_iter pattern: max_iter=1, shared history, warm start (cross_validate.jl:164-175)
        L.fit_b(g, p, ch=ch, verbose=False)
        L.fit_b(gc, p, ch=chc, verbose=False, engine=O.oracle_api())
    assert len(ch.objective) == 10
    assert cases.rel_err(ch.objective, chc.objective) < TOL and cases.fro_err(g.X, gc.X) < TOL
    assert L.objective(g) == pytest.approx(L.objective(gc, engine=O.oracle_api()), rel=1e-10)
    g.close()


def test_error_codes_on_device():
    rng = np.random.default_rng(37)
    pa, X0, Y0 = random_problem(rng, 20, 10, 3, 0.5)
    api = hip()
    bad = _capi.ProblemArrays(pa.m, pa.n, 200, pa.rowptr, pa.colidx, pa.rowvals, pa.colptr, pa.rowidx, pa.colvals, pa.losses, pa.rx, pa.ry)
    with pytest.raises(_capi.GLRMError) as ei:
        api.create(bad)
    assert ei.value.code == _capi.ERR_UNSUPPORTED
    pa.colvals = pa.colvals.copy(); pa.colvals[0] = np.nan
    with pytest.raises(_capi.GLRMError) as ei:
        api.create(pa)
    assert ei.value.code == _capi.ERR_NONFINITE
    pa, X0, Y0 = random_problem(rng, 20, 10, 3, 0.5)
    h = api.create(pa)
    with pytest.raises(_capi.GLRMError) as ei:
        api.fit(h, L.ProxGradParams(max_iter=3), X0, np.zeros_like(Y0))
    assert ei.value.code == _capi.ERR_INVALID
    api.destroy(h)


@pytest.mark.parametrize("mode,mixed", [(1, False), (2, False), (2, True)])
def test_two_shards_on_one_gpu_equal_one_shard(mode, mixed):
    """Sharding only re-labels which handle runs an independent row / column: two shard handles bound to the
    same device buffers, stepped one after the other, give the single-handle bits.
    mixed: a loss per column -- the tiled column passes then read their tiles with the chunk walk i ^ p, p from the GLOBAL column id
    (glrm_tiled.hpp: tile_rot); the second shard starts at column 50, off the eight-column pattern, and must still give the same bits."""
    import torch
    rng = np.random.default_rng(38)
    m, n, k = 500, 120, 32
    losses = [[L.QuadLoss(), L.HuberLoss(), L.L1Loss(0.7), L.QuantileLoss(1.0, quantile=0.3)][j % 4] for j in range(n)] if mixed else None
    pa, X0, Y0 = random_problem(rng, m, n, k, 0.3, losses=losses)
    api = hip()
    params = L.ProxGradParams(max_iter=6)
    o1, X1, Y1, st1 = cases.run_engine(api, pa, X0, Y0, params, tiled=mode)
    assert st1["tiled"] == ((3 | 512 | 256) if mode == 2 else 0)   # LDS tiles, at rank 32 in their lane-per-segment form
    def shard(rb, re, cb, ce):
        r0, r1, c0, c1 = pa.rowptr[rb], pa.rowptr[re], pa.colptr[cb], pa.colptr[ce]
        return _capi.ProblemArrays(m, n, k, np.ascontiguousarray(pa.rowptr[rb:re + 1] - r0), np.ascontiguousarray(pa.colidx[r0:r1]),
                                   np.ascontiguousarray(pa.rowvals[r0:r1]), np.ascontiguousarray(pa.colptr[cb:ce + 1] - c0),
                                   np.ascontiguousarray(pa.rowidx[c0:c1]), np.ascontiguousarray(pa.colvals[c0:c1]),
                                   pa.losses, pa.rx, pa.ry, rb, re, cb, ce)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    hs = [api.create(shard(0, 230, 0, 50), stream=stream, tiled=mode), api.create(shard(230, m, 50, n), stream=stream, tiled=mode)]
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


def test_runs_are_bitwise_deterministic():
    rng = np.random.default_rng(39)
    pa, X0, Y0 = random_problem(rng, 300, 200, 64, 0.2)
    params = L.ProxGradParams(max_iter=8)
    for mode in (1, 2):
        a = cases.run_engine(hip(), pa, X0, Y0, params, tiled=mode)
        b = cases.run_engine(hip(), pa, X0, Y0, params, tiled=mode)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("mode", [1, 2])
def test_step_x_range_chunks_equal_full_sweep(mode):
    """glrm_hip_step_x_range over consecutive row chunks == one glrm_hip_step_x (what the pipelined multi-GPU host relies on)."""
    rng = np.random.default_rng(90 + mode)
    pa, X0, Y0 = random_problem(rng, 700, 150, 32, 0.3, rx=L.NonNegConstraint())
    api = hip()
    res = []
    for chunks in (None, [(0, 100), (100, 101), (101, 512), (512, 700)]):
        h = api.create(pa, tiled=mode)
        api.set_factors(h, X0, Y0)
        api.reset_stepsizes(h, 1.0)
        for _ in range(3):
            if chunks is None:
                api.step_x(h, 0.01)
            else:
                for b, e in chunks:
                    api.step_x_range(h, b, e, 0.01)
            api.step_y(h, 0.01)
        X, Y = np.zeros_like(X0), np.zeros_like(Y0)
        api.get_factors(h, X, Y)
        st = api.kernel_stats(h)
        api.destroy(h)
        res.append((X, Y, st["trials_x"], st["accepts_x"]))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2:] == res[1][2:]


def test_set_regularizers_keeps_the_handle_and_matches_a_fresh_one():
    rng = np.random.default_rng(95)
    pa, X0, Y0 = random_problem(rng, 200, 90, 12, 0.4, rx=L.QuadReg(1.0), ry=L.QuadReg(1.0))
    api, params = hip(), L.ProxGradParams(max_iter=8)
    h = api.create(pa)
    X, Y = X0.copy(order="F"), Y0.copy(order="F")
    api.fit(h, params, X, Y)
    new = np.array([(1, 0, 0.2)], dtype=_capi.REG_DTYPE)
    api.set_regularizers(h, new, new)
    o2, _ = api.fit(h, params, X, Y)
    api.destroy(h)
    pb = _capi.ProblemArrays(pa.m, pa.n, pa.k, pa.rowptr, pa.colidx, pa.rowvals, pa.colptr, pa.rowidx, pa.colvals, pa.losses, new, new)
    h = api.create(pa)
    Xr, Yr = X0.copy(order="F"), Y0.copy(order="F")
    api.fit(h, params, Xr, Yr)
    api.destroy(h)
    h = api.create(pb)
    o3, _ = api.fit(h, params, Xr, Yr)
    api.destroy(h)
    assert np.array_equal(o2, o3) and np.array_equal(X, Xr) and np.array_equal(Y, Yr)
    with pytest.raises(_capi.GLRMError):
        h = api.create(pa)
        try:
            api.set_regularizers(h, np.repeat(new, 3), new)
        finally:
            api.destroy(h)


def test_skewed_lengths_split_launch():
    """A few very long segments next to many short ones: the gather sweeps give the long ones an 8-wave launch of their own on a
    side stream (fork / join by events; also inside the captured hipGraph of glrm_hip_fit)."""
    rng = np.random.default_rng(90)
    m, n, k = 6000, 60, 8
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) / np.sqrt(k) + 0.1 * rng.standard_normal((m, n))
    mask = rng.random((m, n)) < 0.02
    mask[:, 0] = True                      # one fully observed column: 6000 entries against a mean of ~220
    mask[5, :] = True                      # and one fully observed row
    I, J = np.nonzero(mask)
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.NonNegConstraint(), k, obs=(I, J), X=X0, Y=Y0)
    compare(g.problem_arrays(), np.asfortranarray(X0), np.asfortranarray(Y0), L.ProxGradParams(max_iter=15), tiled=1)


# ---------------------------------------------------------------------------------------------------------------------------
# cached gather row sweep (csrc/glrm_cached.hip): the row's opposing vectors are gathered once into LDS, all passes read them there

def _cached(monkeypatch, on):
    monkeypatch.setenv("GLRM_HIP_CACHED", "1" if on else "0")


@pytest.mark.parametrize("k", [20, 32, 40, 64])
def test_cached_row_sweep_equals_the_gather_sweep_bit_for_bit(monkeypatch, k):
    """Uniform QuadLoss: same observations per lane group in the same order, same butterfly -- the bits of the one-wave gather sweep."""
    rng = np.random.default_rng(100 + k)
    pa, X0, Y0 = random_problem(rng, 300, 90, k, 0.4, rx=L.NonNegConstraint(), ry=L.QuadReg(0.2), dup=True)
    params = L.ProxGradParams(max_iter=8)
    _cached(monkeypatch, False)
    o0, Xa, Ya, st0 = cases.run_engine(hip(), pa, X0, Y0, params, tiled=1, waves_row=1)
    _cached(monkeypatch, True)
    for regs in ("1", "0"):  # the row's vectors in registers / in LDS
        monkeypatch.setenv("GLRM_HIP_CACHED_REGS", regs)
        monkeypatch.setenv("GLRM_HIP_CACHED_WAVES", "1")
        o1, Xb, Yb, st1 = cases.run_engine(hip(), pa, X0, Y0, params, tiled=1, waves_row=1)
        assert not st0["tiled"] & 64 and st1["tiled"] & 64
        assert np.array_equal(o0, o1) and np.array_equal(Xa, Xb) and np.array_equal(Ya, Yb)
        assert st0["trials_x"] == st1["trials_x"] and st0["accepts_x"] == st1["accepts_x"]
    # two waves per row (the default for rows of more than four trips): the wave totals are added in wave order -- rounding only
    monkeypatch.setenv("GLRM_HIP_CACHED_REGS", "1")
    monkeypatch.setenv("GLRM_HIP_CACHED_WAVES", "2")
    o2, Xc, Yc, st2 = cases.run_engine(hip(), pa, X0, Y0, params, tiled=1, waves_row=1)
    assert st2["tiled"] & 64 and len(o2) == len(o0)
    assert cases.rel_err(o2, o0) < 1e-10 and cases.fro_err(Xc, Xa) < 1e-10 and cases.fro_err(Yc, Ya) < 1e-10


def test_cached_row_sweep_against_the_oracle_on_mixed_losses_and_regularizers(monkeypatch):
    _cached(monkeypatch, True)
    rng = np.random.default_rng(77)
    m, n, k = 260, 120, 32
    losses = [[L.QuadLoss(), L.HuberLoss(), L.LogisticLoss(), L.L1Loss(0.5), L.PoissonLoss(), L.OrdinalHingeLoss(1, 6)][j % 6] for j in range(n)]
    doms = [L.default_domain(l) for l in losses]
    Z = rng.standard_normal((m, 4)) @ rng.standard_normal((4, n))
    A = np.array([[L.impute_entry(doms[j], losses[j], Z[i, j]) for j in range(n)] for i in range(m)], dtype=np.float64)
    I, J_ = np.nonzero(rng.random((m, n)) < 0.5)
    X0, Y0 = 0.3 * rng.standard_normal((k, m)), 0.3 * rng.standard_normal((k, n))
    rx = [[L.QuadReg(0.3), L.OneReg(0.2), L.NonNegConstraint(), L.ZeroReg()][i % 4] for i in range(m)]
    g = L.GLRM(A, losses, rx, L.QuadReg(0.1), k, obs=(I, J_), X=X0, Y=Y0)
    pa = g.problem_arrays()
    O.set_threads(4)
    params = L.ProxGradParams(max_iter=10)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, np.asfortranarray(X0), np.asfortranarray(Y0), params)
    o_g, X_g, Y_g, st_g = cases.run_engine(hip(), pa, np.asfortranarray(X0), np.asfortranarray(Y0), params, tiled=1)
    assert st_g["tiled"] & 64 and len(o_g) == len(o_c)
    assert cases.rel_err(o_g, o_c) < TOL and cases.fro_err(X_g, X_c) < TOL and cases.fro_err(Y_g, Y_c) < TOL


def test_cached_row_sweep_in_row_chunks_sparse_solver_and_rows_too_long(monkeypatch):
    _cached(monkeypatch, True)
    rng = np.random.default_rng(5)
    pa, X0, Y0 = random_problem(rng, 240, 64, 32, 0.5)
    api = hip()
    # (a) glrm_hip_step_x_range in chunks = one full sweep
    import torch
    res = []
    for chunks in (1, 3):
        h = api.create(pa, tiled=1)
        assert api.kernel_stats(h)["tiled"] & 64
        api.set_factors(h, X0, Y0)
        api.reset_stepsizes(h, 1.0)
        for _ in range(3):
            if chunks == 1:
                api.step_x(h, 0.01)
            else:
                for c in range(chunks):
                    api.step_x_range(h, c * 80, (c + 1) * 80, 0.01)
            api.step_y(h, 0.01)
        X, Y = np.zeros_like(X0), np.zeros_like(Y0)
        api.get_factors(h, X, Y)
        api.destroy(h)
        res.append((X, Y))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    # (b) SparseProxGradParams (fixed step, no line search) against the oracle
    sp = L.SparseProxGradParams(max_iter=12)
    outs = []
    for a_ in (O.oracle_api(), api):
        h = a_.create(pa, tiled=1) if a_ is api else a_.create(pa)
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        obj, _ = a_.fit_sparse(h, sp, X, Y)
        a_.destroy(h)
        outs.append((np.array(obj), X, Y))
    assert len(outs[0][0]) == len(outs[1][0]) and cases.rel_err(outs[1][0], outs[0][0]) < TOL and cases.fro_err(outs[1][1], outs[0][1]) < TOL
    # (c) a row longer than the LDS budget (160 KB = 640 vectors at k = 32 ... a fully observed 700-column row): falls back
    pa2, X2, Y2 = random_problem(rng, 40, 700, 32, 1.1)
    h = api.create(pa2, tiled=1)
    assert not api.kernel_stats(h)["tiled"] & 64
    api.destroy(h)
