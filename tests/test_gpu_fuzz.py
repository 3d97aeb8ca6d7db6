"""-m gpu: randomized models across the whole descriptor table (14 losses, 5 regularizers, 4 wrappers, duplicates, unsorted lists,
inner iterations, both solvers) -- HIP engine vs CPU oracle through the C ABI, north-star tolerance 1e-5."""
import os

import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi

pytestmark = pytest.mark.gpu
TOL = 1e-5
# Honest bookkeeping: how many iterations every seed really compared, or why it compared none.  The accounting
# tests at the end of the file print the tally and fail when too few seeds compared a meaningful stretch.
PLAIN_SEEDS, ROTATED_SEEDS = range(48), range(5000, 5044)
TALLY = {"plain": {}, "rotated": {}}


def random_loss(rng, allow_vector, exact_only=False):
    kinds = ["quad", "l1", "huber", "quantile", "periodic", "poisson", "ordhinge", "logistic", "whinge"]
    if exact_only:  # the losses the engine and the oracle evaluate with the same instructions (no exp / log / sin)
        kinds = ["quad", "l1", "huber", "quantile", "ordhinge", "whinge"]
    if allow_vector:
        kinds += ["mnl", "ova", "ovah", "bvs", "bvsh", "ordistic", "mnlord"]
    kind = kinds[int(rng.integers(len(kinds)))]
    s = float(0.5 + rng.random())
    d = int(rng.integers(2, 7))
    return {
        "quad": lambda: L.QuadLoss(s), "l1": lambda: L.L1Loss(s), "huber": lambda: L.HuberLoss(s, crossover=float(0.5 + rng.random())),
        "quantile": lambda: L.QuantileLoss(s, quantile=float(0.2 + 0.6 * rng.random())), "periodic": lambda: L.PeriodicLoss(float(1 + 3 * rng.random()), s),
        "poisson": lambda: L.PoissonLoss(20), "ordhinge": lambda: L.OrdinalHingeLoss(1, int(rng.integers(3, 9)), s), "logistic": lambda: L.LogisticLoss(s),
        "whinge": lambda: L.WeightedHingeLoss(s, case_weight_ratio=float(0.5 + 2 * rng.random())), "mnl": lambda: L.MultinomialLoss(d, s),
        "ova": lambda: L.OvALoss(d, s), "ovah": lambda: L.OvALoss(d, s, bin_loss=L.HingeLoss(s)), "bvs": lambda: L.BvSLoss(d + 1, s),
        "bvsh": lambda: L.BvSLoss(d + 1, s, bin_loss=L.HingeLoss()), "ordistic": lambda: L.OrdisticLoss(d, s), "mnlord": lambda: L.MultinomialOrdinalLoss(d + 1, s),
    }[kind]()


def column_data(rng, lo, z):
    if hasattr(lo, "max") and lo.embedding_dim > 1:
        return cases._levels(z, 1, lo.max)
    if isinstance(lo, L.OrdinalHingeLoss):
        return cases._levels(z, lo.min, lo.max)
    if isinstance(lo, L.PoissonLoss):
        return np.minimum(np.round(np.exp(z / 2)), 20)
    if isinstance(lo, L.PeriodicLoss):
        return np.mod(z, lo.T)
    if lo.classification:
        return (z > 0).astype(float)
    return z


def random_reg(rng):
    return [L.QuadReg(float(0.05 + rng.random())), L.OneReg(float(0.02 + 0.2 * rng.random())), L.ZeroReg(), L.NonNegConstraint(),
            L.QuadReg(0.1)][int(rng.integers(5))]


def random_model(seed, exact_scalar=False):
    """exact_scalar: a scalar-loss model of the losses both sides evaluate with the same instructions, long run (the reference-order leg)."""
    rng = np.random.default_rng(seed)
    m, n = int(rng.integers(20, 400)), int(rng.integers(5, 60))
    k = int([1, 2, 3, 5, 8, 9, 16, 20, 33, 64][int(rng.integers(10))])
    vector = bool(rng.random() < 0.6) and not exact_scalar
    if vector:
        k = min(k, 33)
    losses = [random_loss(rng, vector, exact_only=exact_scalar) for _ in range(n)]
    offset = vector and rng.random() < 0.5 and k >= 2
    if rng.random() < 0.3:
        rx = [random_reg(rng) for _ in range(m)]
    else:
        rx = random_reg(rng)
    ry = []
    for lo in losses:
        r = random_reg(rng)
        if offset and isinstance(lo, (L.MultinomialOrdinalLoss,)):
            r = L.MNLOrdinalReg(r)
        elif offset and isinstance(lo, (L.BvSLoss, L.OrdisticLoss)) and rng.random() < 0.7:
            r = L.OrdinalReg(r)
        ry.append(r)
    Z = rng.standard_normal((m, 3))
    A = np.column_stack([column_data(rng, lo, Z @ rng.standard_normal(3)) for lo in losses])
    style = int(rng.integers(3))
    kw = {}
    if style == 0:      # sorted lists from a mask
        I, J = np.nonzero(rng.random((m, n)) < 0.2 + 0.6 * rng.random())
        kw["obs"] = (I, J)
    elif style == 1:    # sampled with replacement: duplicates, unsorted
        nobs = int(m * n * 0.3)
        kw["obs"] = (rng.integers(0, m, nobs), rng.integers(0, n, nobs))
    D = L.embedding_dim(losses)
    # u = x'y ~ N(0, 1) at the start: keeps exp(u) of the Poisson / logistic columns in range
    g = L.GLRM(A, losses, rx, ry, k, X=rng.standard_normal((k, m)) / k ** 0.25, Y=rng.standard_normal((k, D)) / k ** 0.25, offset=offset, **kw)
    inner = int(rng.integers(1, 4)) if rng.random() < 0.3 else 1
    p = L.ProxGradParams(float([1.0, 0.5, 2.0][int(rng.integers(3))]), max_iter=int(rng.integers(40, 120) if exact_scalar else rng.integers(4, 14)), inner_iter=inner)
    return g, p


def well_conditioned_prefix(pa, X0, Y0, p, seed):
    """Non-smooth losses, strict `<` decisions and exploding starts make some random trajectories amplify rounding by orders of
    magnitude per iteration (the oracle itself, started from X0 * (1 + 1e-13 * noise), then drifts apart just as fast).  Parity
    is only defined while the trajectory is stable: return the largest iteration count T for which the perturbed oracle run stays
    within 1e-9 of the unperturbed one on every recorded objective and within 1e-8 on the factors."""
    rng = np.random.default_rng(10_000 + seed)
    api = O.oracle_api()
    Xp = X0 * (1 + 1e-13 * rng.standard_normal(X0.shape))
    T = p.max_iter
    while T >= 2:
        q = L.ProxGradParams(p.stepsize, max_iter=T, inner_iter=p.inner_iter_X, abs_tol=0.0, rel_tol=-1.0)
        o_a, X_a, Y_a, _ = cases.run_engine(api, pa, X0, Y0, q)
        o_b, X_b, Y_b, _ = cases.run_engine(api, pa, Xp, Y0, q)
        if len(o_a) != len(o_b):
            # one of the two runs stopped early (with these tolerances: the recorded objective ROSE after iteration 10,
            # src/algorithms/proxgrad.jl:210-213) and the other did not -- the stop decision at that iteration hangs on the perturbation:
            # the stable prefix ends before it (soak seed 30301: 13 vs 14 recorded objectives)
            T = min(len(o_a), len(o_b)) - 2
            continue
        with np.errstate(all="ignore"):
            rel = np.abs(o_a - o_b) / np.abs(o_a)
        ok = (rel < 1e-9) | (o_a == o_b)
        bad = np.flatnonzero(~ok[1:])
        if len(bad):
            T = int(bad[0])  # iterations 1..bad[0] were fine
            continue
        if cases.fro_err(X_b, X_a) < 1e-8 and cases.fro_err(Y_b, Y_a) < 1e-8:
            return T
        T -= max(1, T // 3)
    return 0


@pytest.mark.parametrize("seed", PLAIN_SEEDS)
def test_random_models_match_oracle(seed):
    g, p = random_model(seed)
    pa = g.problem_arrays()
    X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
    O.set_threads(4)
    stable = well_conditioned_prefix(pa, X0, Y0, p, seed)
    TALLY["plain"][seed] = ("skipped: unstable within two iterations", 0)
    if stable < 2:
        pytest.skip("trajectory amplifies a 1e-13 perturbation beyond 1e-9 within two iterations")
    p = L.ProxGradParams(p.stepsize, max_iter=min(p.max_iter, stable), inner_iter=p.inner_iter_X, abs_tol=0.0, rel_tol=-1.0)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, p)
    o_g, X_g, Y_g, st_g = cases.run_engine(_capi.hip_api(), pa, X0, Y0, p)
    assert len(o_g) == len(o_c)
    assert cases.rel_err(o_g, o_c) < TOL and cases.fro_err(X_g, X_c) < TOL and cases.fro_err(Y_g, Y_c) < TOL, (
        seed, stable, cases.rel_err(o_g, o_c), cases.fro_err(X_g, X_c), cases.fro_err(Y_g, Y_c))
    assert st_g["nnz_rows"] == st_c["nnz_rows"] and st_g["nnz_cols"] == st_c["nnz_cols"]
    TALLY["plain"][seed] = ("compared", len(o_g) - 1)


@pytest.mark.parametrize("seed", range(100, 110))
def test_random_models_sparse_solver(seed):
    g, _ = random_model(seed)
    pa = g.problem_arrays()
    p = L.SparseProxGradParams(max_iter=15)
    res = []
    for api in (O.oracle_api(), _capi.hip_api()):
        h = api.create(pa)
        try:
            X, Y = np.array(g.X, order="F"), np.array(g.Y, order="F")
            obj, _ = api.fit_sparse(h, p, X, Y)
            res.append((obj, X, Y))
        finally:
            api.destroy(h)
    assert len(res[0][0]) == len(res[1][0])
    assert cases.rel_err(res[1][0], res[0][0]) < TOL and cases.fro_err(res[1][1], res[0][1]) < TOL and cases.fro_err(res[1][2], res[0][2]) < TOL


@pytest.mark.parametrize("seed", ROTATED_SEEDS)
def test_random_models_with_the_sweep_family_rotated(seed):
    """tests/perf/soak_fuzz.py on seeds of its own: the model of the seed on the family the seed selects (auto / gather only / LDS-tiled /
    phase-aligned passes with small super-tiles / cached row sweep, persistent or not) against the oracle, and on every third seed three
    ragged shards on one device against the single handle, bit for bit.  A seed whose trajectory is unstable within two iterations is
    SKIPPED; a deviation is skipped as "ill-conditioned" only if the oracle does not reproduce itself from reversed observation lists or
    1e-13-perturbed starts (profiles/r03_soak_fuzz.txt: 7 of 2 000 seeds) -- with the oracle's own deviation in the skip reason."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("soak_fuzz", os.path.join(os.path.dirname(os.path.abspath(__file__)), "perf", "soak_fuzz.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    O.set_threads(4)
    res, fam, detail = soak.one(seed)
    TALLY["rotated"][seed] = (f"{res} [{fam}]", soak.LAST["iterations"] if res == "ok" else 0)
    if res == "skip":
        pytest.skip(f"[{fam}] trajectory amplifies a 1e-13 perturbation beyond 1e-9 within two iterations")
    if res == "ill-conditioned":
        pytest.skip(f"[{fam}] ill-conditioned, nothing asserted: {detail}")
    assert res == "ok", (seed, fam, res, detail)


EXACT_KINDS = {0, 1, 2, 3, 6, 8}   # Quad, L1, Huber, Quantile, OrdinalHinge, WeightedHinge: the same instructions on both sides (no exp / log / sin)
REFORDER_SEEDS = range(48)
REFORDER_TALLY = {}


@pytest.mark.parametrize("seed", REFORDER_SEEDS)
def test_random_models_over_their_whole_trajectory_in_the_reference_order_mode(seed):
    """the randomized evidence compared only the stable prefix of each trajectory (median 7 iterations), because another
    summation order forks an unstable trajectory.  In the reference-order mode (glrm_options.sum_order = 1) there is no other order: the
    engine adds like the oracle, so the WHOLE run of every scalar-loss seed is compared -- every iteration the solver takes, unstable or
    not, under the seed's own stop rule -- and must be IDENTICAL, bit for bit, when the model holds only losses both sides evaluate with
    the same instructions; models with Logistic / Poisson / Periodic columns (in-kernel exp / log / sin vs libm) are held to 1e-9 on
    their stable prefix.  Models outside the mode (multi-dimensional losses, offsets, k > 64) are counted, not compared."""
    g, p = random_model(seed, exact_scalar=seed % 2 == 0)   # even seeds: exactly evaluated losses, 40-120 iterations; odd seeds: the plain table
    pa = g.problem_arrays()
    scalar = L.embedding_dim(g.losses) == g.n and not getattr(g, "offset", False) and all(int(r["wrap"]) == 0 for r in list(pa.rx) + list(pa.ry))
    if not scalar or g.k > 64:
        REFORDER_TALLY[seed] = ("outside the mode", 0)
        pytest.skip("multi-dimensional losses / wrapped regularizers / k > 64: outside the reference-order mode")
    X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
    exact = all(int(l["kind"]) in EXACT_KINDS for l in pa.losses)
    O.set_threads(4)
    if not exact:
        stable = well_conditioned_prefix(pa, X0, Y0, p, seed)
        if stable < 2:
            REFORDER_TALLY[seed] = ("transcendental, unstable", 0)
            pytest.skip("exp / log / sin based losses on a trajectory that amplifies 1e-13 within two iterations")
        p = L.ProxGradParams(p.stepsize, max_iter=min(p.max_iter, stable), inner_iter=p.inner_iter_X, abs_tol=0.0, rel_tol=-1.0)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, p)
    o_g, X_g, Y_g, st_g = cases.run_engine(_capi.hip_api(), pa, X0, Y0, p, sum_order=1)
    assert st_g["tiled"] == 128 and len(o_g) == len(o_c), (seed, st_g["tiled"], len(o_g), len(o_c))
    if exact:
        same_obj = np.array_equal(o_g[1:], o_c[1:]) or (np.isnan(o_g[1:]) == np.isnan(o_c[1:])).all() and np.array_equal(np.nan_to_num(o_g[1:]), np.nan_to_num(o_c[1:]))
        assert same_obj and np.array_equal(X_g, X_c) and np.array_equal(Y_g, Y_c), (seed, cases.rel_err(o_g[1:], o_c[1:]), np.abs(X_g - X_c).max(), np.abs(Y_g - Y_c).max())
        for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
            assert st_g[key] == st_c[key], (seed, key)
        REFORDER_TALLY[seed] = ("bit-identical, whole run", len(o_g) - 1)
    else:
        assert cases.rel_err(o_g, o_c) < 1e-9 and cases.fro_err(X_g, X_c) < 1e-9 and cases.fro_err(Y_g, Y_c) < 1e-9, seed
        REFORDER_TALLY[seed] = ("1e-9, stable prefix", len(o_g) - 1)


def test_accounting_reference_order_seeds(capsys):
    t = REFORDER_TALLY
    if len(t) < len(REFORDER_SEEDS):
        pytest.skip("select the whole file for the accounting")
    whole = [v[1] for v in t.values() if v[0].startswith("bit")]
    line = (f"[fuzz accounting] reference-order mode: {len(t)} seeds, {len(whole)} bit-identical over their WHOLE run (iterations: min "
            f"{min(whole, default=0)}, median {int(np.median(whole or [0]))}, max {max(whole, default=0)}), "
            f"{sum(1 for v in t.values() if v[0].startswith('1e-9'))} with exp / log / sin losses at 1e-9 on the stable prefix, "
            f"{sum(1 for v in t.values() if v[0].startswith('outside'))} outside the mode, "
            f"{sum(1 for v in t.values() if v[0].startswith('transc'))} transcendental and unstable")
    with capsys.disabled():
        print("\n" + line)
    assert len(whole) >= 20 and np.median(whole) >= 11, line


def _account(kind, seeds, need):
    t = TALLY[kind]
    if len(t) < len(seeds):
        pytest.skip(f"only {len(t)} of {len(seeds)} {kind} seeds ran in this session (select the whole file for the accounting)")
    its = [t[s_][1] for s_ in seeds]
    done = [v for v in its if v > 0]
    long_ = sum(1 for v in its if v >= 4)
    line = (f"[fuzz accounting] {kind}: {len(seeds)} seeds, {len(done)} compared (iterations: min {min(done, default=0)}, "
            f"median {int(np.median(done or [0]))}, max {max(its)}), {long_} compared >= 4 iterations, "
            f"{sum(1 for s_ in seeds if t[s_][0].startswith('skip'))} unstable within two iterations, "
            f"{sum(1 for s_ in seeds if t[s_][0].startswith('ill'))} ill-conditioned; per seed: {its}")
    return line, long_ >= need


def test_accounting_plain_seeds(capsys):
    """At least 34 of the plain seeds must have compared >= 4 iterations of the trajectory."""
    line, ok = _account("plain", list(PLAIN_SEEDS), 34)
    with capsys.disabled():
        print("\n" + line)
    assert ok, line


def test_accounting_rotated_seeds(capsys):
    """At least 30 of the rotated-family seeds must have compared >= 4 iterations."""
    line, ok = _account("rotated", list(ROTATED_SEEDS), 30)
    with capsys.disabled():
        print("\n" + line)
    assert ok, line


# ------------------------------------------------------------------------------------------------ every sweep family x {Logistic, OrdinalHinge}

FAMILY_ENV = {
    "gather": ({"GLRM_HIP_CACHED": "0", "GLRM_HIP_BLOCKED": "0"}, {"tiled": 1}, 0),
    "tiled": ({}, {"tiled": 2}, 3),
    "blocked": ({"GLRM_HIP_BLOCKED": "3", "GLRM_HIP_BLOCKED_TPS": "1", "GLRM_HIP_BLOCKED_FILL": "3", "GLRM_HIP_CACHED": "0"}, {"tiled": 1}, 48),
    "cached": ({"GLRM_HIP_CACHED": "1", "GLRM_HIP_BLOCKED": "0"}, {"tiled": 0}, 64),
    "cached_nopersist": ({"GLRM_HIP_CACHED": "1", "GLRM_HIP_CACHED_PERSIST": "0", "GLRM_HIP_BLOCKED": "0"}, {"tiled": 0}, 64),
}


@pytest.mark.parametrize("loss", ["logistic", "ordinal_hinge", "mixed"])
@pytest.mark.parametrize("family", list(FAMILY_ENV))
def test_every_sweep_family_on_logistic_and_ordinal_hinge(monkeypatch, family, loss):
    """The five sweep families of the scalar-loss path on the two non-quadratic losses BASELINE config 5 names (src/losses.jl:247-311), one
    descriptor for the whole model (the per-segment kernels) and the Quad / Logistic / OrdinalHinge column mix (a descriptor per column:
    the per-observation row kernels) -- explicitly, not by seed luck.  Sorted lists, 12 iterations from a small start, 1e-5."""
    env, kw, want = FAMILY_ENV[family]
    for k_ in ("GLRM_HIP_BLOCKED", "GLRM_HIP_BLOCKED_TPS", "GLRM_HIP_BLOCKED_FILL", "GLRM_HIP_CACHED", "GLRM_HIP_CACHED_PERSIST"):
        monkeypatch.delenv(k_, raising=False)
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    m, n, k, q = 3000, 1200, 32, 120
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=0, loss_mix=1 if loss == "mixed" else 0)
    if loss == "mixed":
        kinds = [L.QuadLoss().descriptor(), L.LogisticLoss().descriptor(), L.OrdinalHingeLoss(1, 5).descriptor()]
        losses = np.array([kinds[f % 3] for f in range(n)], dtype=_capi.LOSS_DTYPE)
    elif loss == "logistic":
        rowvals, colvals = (rowvals > 0).astype(np.float64), (colvals > 0).astype(np.float64)
        losses = np.array([L.LogisticLoss(1.3).descriptor()], dtype=_capi.LOSS_DTYPE)
    else:
        f = lambda v: np.clip(np.round(3 + 1.5 * v), 1, 5)  # noqa: E731
        rowvals, colvals = f(rowvals), f(colvals)
        losses = np.array([L.OrdinalHingeLoss(1, 5, 0.7).descriptor()], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, losses, reg, reg)
    X0, Y0 = np.asfortranarray(0.3 * X0), np.asfortranarray(0.3 * Y0)
    p = L.ProxGradParams(max_iter=12, abs_tol=0.0, rel_tol=-1.0)
    O.set_threads(4)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, p)
    o_g, X_g, Y_g, st_g = cases.run_engine(_capi.hip_api(), pa, X0, Y0, p, **kw)
    assert st_g["tiled"] & want == want and (want or not st_g["tiled"] & (1 | 2 | 16 | 32 | 64)), (family, st_g["tiled"])
    e = (cases.rel_err(o_g, o_c), cases.fro_err(X_g, X_c), cases.fro_err(Y_g, Y_c))
    assert len(o_g) == len(o_c) and max(e) < TOL, (family, loss, e)
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert abs(st_g[key] - st_c[key]) <= max(5, 0.03 * st_c[key]), (key, st_g[key], st_c[key])
