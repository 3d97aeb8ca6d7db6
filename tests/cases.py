"""Seeded parity cases shared by the CPU tests (oracle vs golden fixtures) and the -m gpu tests
(HIP engine vs oracle vs golden fixtures).  The first four are the fixtures SURVEY.md section 8(c)
asks for; inputs are stored in the fixtures themselves, so they do not depend on numpy's RNG."""
import numpy as np

import lowrankmodels.jl_amd as L


def case_c1(rng):
    """BASELINE config 1: 100x100 dense, rank 5, QuadLoss + QuadReg(.1), default ProxGradParams
    (examples/simple_glrms.jl:31-40 fit_pca_nucnorm shape)."""
    m = n = 100
    k = 5
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n))
    return dict(A=A, losses=L.QuadLoss(), rx=L.QuadReg(.1), ry=L.QuadReg(.1), k=k,
                X=rng.standard_normal((k, m)), Y=rng.standard_normal((k, n))), L.ProxGradParams()


def case_nnmf(rng):
    """60x40 rank 4, 30 % observed, NonNegConstraint on both factors: the initial objective is Inf
    (negative entries in randn factors), the first accepted steps project onto the orthant."""
    m, n, k = 60, 40, 4
    A = rng.random((m, k)) @ rng.random((k, n))
    I, J = np.nonzero(rng.random((m, n)) < 0.3)
    return dict(A=A, losses=L.QuadLoss(), rx=L.NonNegConstraint(), ry=L.NonNegConstraint(), k=k, obs=(I, J),
                X=rng.standard_normal((k, m)), Y=rng.standard_normal((k, n))), L.ProxGradParams(max_iter=60)


def case_mixed(rng):
    """50x30 rank 3, heterogeneous columns (Quad / Logistic / OrdinalHinge(1,5) by f mod 3, random scales),
    per-row regularizers, Omega sampled WITH replacement (duplicates, test/hello_world.jl:48) and the
    row / column views deliberately different (examples/censored.jl:26-27 passes both lists)."""
    m, n, k = 50, 30, 3
    Z = rng.standard_normal((m, k)) @ rng.standard_normal((k, n))
    A = np.zeros((m, n))
    losses = []
    for f in range(n):
        s = float(0.5 + rng.random())
        if f % 3 == 0:
            A[:, f] = Z[:, f] + 0.1 * rng.standard_normal(m)
            losses.append(L.QuadLoss(s))
        elif f % 3 == 1:
            A[:, f] = (rng.random(m) < 1 / (1 + np.exp(-Z[:, f]))).astype(float)
            losses.append(L.LogisticLoss(s))
        else:
            A[:, f] = np.clip(np.round(3 + 1.5 * Z[:, f]), 1, 5)
            losses.append(L.OrdinalHingeLoss(1, 5, s))
    rx = [L.QuadReg(), L.OneReg(0.5), L.NonNegConstraint(), L.ZeroReg()] + [L.QuadReg(0.3) for _ in range(m - 4)]
    nobs = 5 * max(m, n)
    feats = [[] for _ in range(m)]
    exs = [[] for _ in range(n)]
    for i, j in zip(rng.integers(0, m, nobs), rng.integers(0, n, nobs)):
        feats[i].append(int(j))
    for i, j in zip(rng.integers(0, m, nobs), rng.integers(0, n, nobs)):
        exs[j].append(int(i))
    return dict(A=A, losses=losses, rx=rx, ry=L.QuadReg(0.2), k=k, observed_features=feats, observed_examples=exs,
                X=rng.standard_normal((k, m)), Y=rng.standard_normal((k, n))), L.ProxGradParams(max_iter=40)


def case_kmeans(rng):
    """40x40 rank 6, UnitOneSparseConstraint on X (k-means), ZeroReg on Y, inner_iter=10
    (examples/simple_glrms.jl:43-57 fit_kmeans; test/runtests.jl:19-25 uses inner_iter=10)."""
    m, n, k = 40, 40, 6
    C = rng.standard_normal((k, n)) * 3
    A = C[rng.integers(0, k, m)] + 0.1 * rng.standard_normal((m, n))
    return dict(A=A, losses=L.QuadLoss(), rx=L.UnitOneSparseConstraint(), ry=L.ZeroReg(), k=k,
                X=rng.standard_normal((k, m)), Y=rng.standard_normal((k, n))), L.ProxGradParams(max_iter=15, inner_iter=10)


def case_c4(rng):
    """BASELINE config 4 (the north-star target) at 1/60000 of the size: rank 64, QuadLoss, NonNegConstraint on X and Y
    (src/regularizers.jl:101-114), every row observes q sorted columns (one per stratum, like the synthetic generator), values from
    non-negative factors.  Two starts in the suite: this fixture starts from |N(0,1)|/sqrt(k) (the start bench.py uses at C4);
    the N(0,1) default start of an NNMF of this shape collapses to X = 0 (tests/test_gpu_parity.py::test_c4_recipe covers both)."""
    m, n, k, q = 150, 80, 64, 10
    S = n // q
    A = (rng.random((m, k)) / np.sqrt(k)) @ (rng.random((k, n)) / np.sqrt(k)) + 0.01 * rng.standard_normal((m, n))
    I = np.repeat(np.arange(m), q)
    J = (np.arange(q)[None, :] * S + rng.integers(0, S, (m, q))).ravel()
    return dict(A=A, losses=L.QuadLoss(), rx=L.NonNegConstraint(), ry=L.NonNegConstraint(), k=k, obs=(I, J),
                X=np.abs(rng.standard_normal((k, m))) / np.sqrt(k), Y=np.abs(rng.standard_normal((k, n))) / np.sqrt(k)), L.ProxGradParams(max_iter=30)


def _levels(z, lo, hi):
    return np.clip(np.round((lo + hi) / 2 + z), lo, hi)


def _observe(rng, m, n, density):
    mask = rng.random((m, n)) < density
    return [list(map(int, np.flatnonzero(mask[i]))) for i in range(m)], [list(map(int, np.flatnonzero(mask[:, j]))) for j in range(n)]


def case_loss_test(rng):
    """The model of the reference's test/loss_test.jl:6-69 -- every loss constructor in one GLRM, k = 5, QuadReg(1) on X,
    OrdinalReg(QuadReg(1)) on the multi-dimensional ordinal columns and QuadReg(1) elsewhere, scale=false, offset=false --
    at m = 80 (the reference uses 1000) with 85 % of the entries observed."""
    losses = [L.QuadLoss(), L.QuadLoss(10), L.L1Loss(), L.L1Loss(5.2), L.HuberLoss(), L.HuberLoss(4), L.HuberLoss(3.1, crossover=3.2),
              L.PeriodicLoss(2 * np.pi), L.PeriodicLoss(2 * np.pi, 4), L.PoissonLoss(20), L.PoissonLoss(22), L.OrdinalHingeLoss(1, 10),
              L.OrdinalHingeLoss(2, 7, 5), L.LogisticLoss(), L.LogisticLoss(0.2), L.WeightedHingeLoss(), L.WeightedHingeLoss(11),
              L.WeightedHingeLoss(1.5, case_weight_ratio=4.3), L.MultinomialLoss(4), L.MultinomialLoss(6, .5), L.MultinomialOrdinalLoss(3)]
    m, n, k = 80, len(losses), 5
    D = L.embedding_dim(losses)
    XY = (rng.standard_normal((m, k)) @ rng.standard_normal((k, D)))
    A = np.zeros((m, n))
    for f, (lo, (y0, y1)) in enumerate(zip(losses, L.get_yidxs(losses))):
        z = XY[:, y0]
        if isinstance(lo, L.MultinomialLoss):
            A[:, f] = 1 + np.argmax(XY[:, y0:y1], axis=1)
        elif isinstance(lo, L.MultinomialOrdinalLoss):
            A[:, f] = _levels(z, 1, lo.max)
        elif isinstance(lo, L.OrdinalHingeLoss):
            A[:, f] = _levels(z, lo.min, lo.max)
        elif isinstance(lo, L.PoissonLoss):
            A[:, f] = np.minimum(np.round(np.exp(z / 3)), 20)
        elif isinstance(lo, L.PeriodicLoss):
            A[:, f] = np.mod(z, lo.T)
        elif lo.classification:
            A[:, f] = z > 0
        else:
            A[:, f] = z
    ry = [L.OrdinalReg(L.QuadReg(1)) if isinstance(lo, (L.MultinomialOrdinalLoss, L.OrdisticLoss)) else L.QuadReg(1) for lo in losses]
    feats, exs = _observe(rng, m, n, 0.85)
    return dict(A=A, losses=losses, rx=L.QuadReg(1), ry=ry, k=k, observed_features=feats, observed_examples=exs,
                X=rng.standard_normal((k, m)), Y=rng.standard_normal((k, D))), L.ProxGradParams(max_iter=30)


def case_mnl(rng):
    """test/prob_tests/MultinomialLoss.jl:9-41 at m, n = 60, 12: n categorical columns with K = 4 levels sampled from the
    multinomial logit of a rank-2 model, fitted with rank 3 and QuadReg() on both factors, fully observed."""
    m, n, k, K = 60, 12, 3, 4
    XY = rng.standard_normal((m, 2)) @ rng.standard_normal((2, n * K))
    A = np.zeros((m, n))
    for j in range(n):
        u = XY[:, j * K:(j + 1) * K]
        w = np.exp(-(u - u.mean(axis=1, keepdims=True)))
        w /= w.sum(axis=1, keepdims=True)
        A[:, j] = 1 + (rng.random((m, 1)) > np.cumsum(w, axis=1)).sum(axis=1).clip(0, K - 1)
    return dict(A=A, losses=[L.MultinomialLoss(K) for _ in range(n)], rx=L.QuadReg(), ry=L.QuadReg(), k=k,
                X=rng.standard_normal((k, m)), Y=rng.standard_normal((k, n * K))), L.ProxGradParams(max_iter=40)


def case_mnl_ordinal(rng):
    """test/prob_tests/MultinomialOrdinalLoss.jl:9-58 at m, n = 50, 14: ordinal columns with 5 levels, rank 3 fit with
    lastentry1(QuadReg(.01)) on X and MNLOrdinalReg(QuadReg(.01)) on Y, and Y started from prox!(ry, Y_block, 1)."""
    m, n, k, nlevels = 50, 14, 3, 5
    d = nlevels - 1
    z = rng.standard_normal((m, 2)) @ rng.standard_normal((2, n))
    A = _levels(1.2 * z, 1, nlevels)
    ry = L.MNLOrdinalReg(L.QuadReg(.01))
    Y = rng.standard_normal((k, n * d))
    for j in range(n):
        Y[:, j * d:(j + 1) * d] = ry.prox(Y[:, j * d:(j + 1) * d], 1)
    return dict(A=A, losses=[L.MultinomialOrdinalLoss(nlevels) for _ in range(n)], rx=L.lastentry1(L.QuadReg(.01)), ry=ry, k=k,
                X=rng.standard_normal((k, m)), Y=Y), L.ProxGradParams(max_iter=40)


def case_categorical_mix(rng):
    """Multinomial / OvA (logistic and hinge bin_loss) / BvS next to scalar columns, partially observed."""
    losses = [L.MultinomialLoss(4), L.QuadLoss(), L.OvALoss(3, bin_loss=L.LogisticLoss()), L.BvSLoss(5),
              L.OvALoss(4, bin_loss=L.HingeLoss()), L.LogisticLoss(), L.BvSLoss(3, bin_loss=L.HingeLoss()), L.MultinomialLoss(3, 0.7)]
    m, k = 28, 3
    return _multidim_data(rng, m, k, losses, L.QuadReg(0.1), L.QuadReg(0.2), L.ProxGradParams(max_iter=12))


def case_ordinal_offsets(rng):
    """The reference's ordinal recipe: lastentry1 on X, OrdinalReg / MNLOrdinalReg / lastentry_unpenalized on Y's blocks."""
    losses = [L.OrdisticLoss(4), L.MultinomialOrdinalLoss(5), L.QuadLoss(), L.BvSLoss(4), L.MultinomialOrdinalLoss(3)]
    ry = [L.lastentry_unpenalized(L.QuadReg(0.1)), L.MNLOrdinalReg(L.QuadReg(0.1)), L.lastentry_unpenalized(L.QuadReg(0.3)),
          L.OrdinalReg(L.QuadReg(0.1)), L.MNLOrdinalReg(L.ZeroReg())]
    return _multidim_data(rng, 28, 3, losses, L.lastentry1(L.QuadReg(0.1)), ry, L.ProxGradParams(max_iter=12))


def case_offsets_scalar(rng):
    """add_offset! on a scalar-loss model (src/modify_glrm.jl:20-25), inner_iter = 2."""
    losses = [L.QuadLoss(), L.HuberLoss(), L.LogisticLoss(), L.QuadLoss(0.5), L.L1Loss()]
    return _multidim_data(rng, 28, 3, losses, L.lastentry1(L.OneReg(0.05)), L.lastentry_unpenalized(L.QuadReg(0.2)),
                          L.ProxGradParams(max_iter=14, inner_iter=2))


def _multidim_data(rng, m, k, losses, rx, ry, params):
    n = len(losses)
    Z = rng.standard_normal((m, k))
    A = np.zeros((m, n))
    for f, lo in enumerate(losses):
        z = Z @ rng.standard_normal(k)
        if hasattr(lo, "max"):
            A[:, f] = _levels(z, 1, lo.max)
        elif lo.classification:
            A[:, f] = z > 0
        else:
            A[:, f] = z
    feats, exs = _observe(rng, m, n, 0.8)
    return dict(A=A, losses=losses, rx=rx, ry=ry, k=k, observed_features=feats, observed_examples=exs,
                X=rng.standard_normal((k, m)), Y=rng.standard_normal((k, L.embedding_dim(losses)))), params


MULTIDIM_CASES = {"categorical_mix": (case_categorical_mix, 21), "ordinal_offsets": (case_ordinal_offsets, 22),
                  "offsets_scalar": (case_offsets_scalar, 23), "mnl": (case_mnl, 24), "mnl_ordinal": (case_mnl_ordinal, 25),
                  "loss_test": (case_loss_test, 26)}


def build_multidim_case(name):
    fn, seed = MULTIDIM_CASES[name]
    return fn(np.random.default_rng(seed))


GOLDEN_CASES = {"c1": (case_c1, 11), "c4": (case_c4, 15), "nnmf": (case_nnmf, 12), "mixed": (case_mixed, 13), "kmeans": (case_kmeans, 14),
                "mnl_ordinal": (case_mnl_ordinal, 25), "loss_test": (case_loss_test, 26)}


def build_golden_case(name):
    fn, seed = GOLDEN_CASES[name]
    return fn(np.random.default_rng(seed))


# ---------------------------------------------------------------------------------- fixtures on disk

def _desc_arrays(objs, kind):
    return np.array([o.descriptor() for o in objs], dtype=float)


def save_case(path, kwargs, params, outputs):
    g = L.GLRM(**kwargs)
    pa = g.problem_arrays()
    np.savez_compressed(
        path, m=g.m, n=g.n, k=g.k,
        rowptr=pa.rowptr, colidx=pa.colidx, rowvals=pa.rowvals, colptr=pa.colptr, rowidx=pa.rowidx, colvals=pa.colvals,
        losses=np.array([l.descriptor() for l in g.losses], dtype=float),
        rx=np.array([r.descriptor() for r in g.rx], dtype=float), ry=np.array([r.descriptor() for r in g.ry], dtype=float),
        X0=kwargs["X"], Y0=kwargs["Y"],
        params=np.array([params.stepsize, params.max_iter, params.inner_iter_X, params.inner_iter_Y, params.abs_tol,
                         params.rel_tol, params.min_stepsize]),
        **outputs)


def load_case(path):
    """-> (ProblemArrays, X0, Y0, params, fixture dict).  Uses only the stored inputs."""
    from lowrankmodels.jl_amd import _capi
    z = np.load(path)

    def pack(desc, dtype, ncol):
        d = z[desc]
        rows = [tuple([int(r[0]), int(r[1])] + [float(v) for v in r[2:ncol]]) for r in d]
        if len(set(rows)) == 1:
            rows = rows[:1]
        return np.array(rows, dtype=dtype)

    pa = _capi.ProblemArrays(int(z["m"]), int(z["n"]), int(z["k"]), z["rowptr"], z["colidx"], z["rowvals"], z["colptr"],
                             z["rowidx"], z["colvals"], pack("losses", _capi.LOSS_DTYPE, 5), pack("rx", _capi.REG_DTYPE, 3),
                             pack("ry", _capi.REG_DTYPE, 3))
    p = z["params"]
    params = L.ProxGradParams(p[0], max_iter=int(p[1]), inner_iter_X=int(p[2]), inner_iter_Y=int(p[3]), abs_tol=p[4],
                              rel_tol=p[5], min_stepsize=p[6])
    return pa, np.asfortranarray(z["X0"]), np.asfortranarray(z["Y0"]), params, z


def run_engine(api, pa, X0, Y0, params, **create_kw):
    """fit through the C ABI (whole-fit entry point); returns objective, X, Y, stats."""
    h = api.create(pa, **create_kw)
    try:
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        obj, sec = api.fit(h, params, X, Y)
        st = api.kernel_stats(h)
    finally:
        api.destroy(h)
    return obj, X, Y, st


def rel_err(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), "finite/non-finite pattern differs"
    assert np.array_equal(a[~fin], b[~fin]) or (np.isnan(a[~fin]) == np.isnan(b[~fin])).all()
    if not fin.any():
        return 0.0
    den = np.maximum(np.abs(b[fin]), 1e-300)
    return float(np.max(np.abs(a[fin] - b[fin]) / den))


def fro_err(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / (nb if nb > 0 else 1.0))


# ---------------------------------------------------------------------------------- sharded set-up (hosts' protocol)

def shard_of(pa, rb, re, cb, ce):
    """Rows [rb, re) and columns [cb, ce) of a whole-problem ProblemArrays as one shard (list problems, host arrays)."""
    from lowrankmodels.jl_amd import _capi
    r0, r1, c0, c1 = int(pa.rowptr[rb]), int(pa.rowptr[re]), int(pa.colptr[cb]), int(pa.colptr[ce])
    rx = pa.rx if len(pa.rx) == 1 else np.ascontiguousarray(pa.rx[rb:re])
    ry = pa.ry if len(pa.ry) == 1 else np.ascontiguousarray(pa.ry[cb:ce])
    return _capi.ProblemArrays(pa.m, pa.n, pa.k, np.ascontiguousarray(pa.rowptr[rb:re + 1] - r0), np.ascontiguousarray(pa.colidx[r0:r1]),
                               np.ascontiguousarray(pa.rowvals[r0:r1]), np.ascontiguousarray(pa.colptr[cb:ce + 1] - c0),
                               np.ascontiguousarray(pa.rowidx[c0:c1]), np.ascontiguousarray(pa.colvals[c0:c1]),
                               pa.losses, rx, ry, rb, re, cb, ce)


def run_shards_on_one_device(api, pa, X0, Y0, params, row_bounds, col_bounds, x_chunks=1, **create_kw):
    """What a sharding host does (include/glrm_hip.h, glrm_signature), with every shard on device 0 and the "exchange" being the
    shared buffers: create each shard with GLRM_PROBLEM_DEFER_SETUP, combine the signatures, finalize, then step all shards through
    the outer iterations.  Returns (objective[1:], X, Y, [kernel_stats per shard])."""
    import torch
    from lowrankmodels.jl_amd import _capi
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    ns = len(row_bounds) - 1
    hs = [api.create(shard_of(pa, row_bounds[s], row_bounds[s + 1], col_bounds[s], col_bounds[s + 1]), stream=stream, defer=True, **create_kw)
          for s in range(ns)]
    try:
        whole = _capi.CSignature.combine([api.signature(h) for h in hs])
        for h in hs:
            api.finalize(h, whole)
        ld = api.factor_ld(hs[0])
        d = pa.d
        dX, dY = torch.zeros(pa.m * ld, dtype=torch.float64, device=dev), torch.zeros(d * ld, dtype=torch.float64, device=dev)
        dC, dR = torch.zeros(pa.n, dtype=torch.float64, device=dev), torch.zeros(pa.m, dtype=torch.float64, device=dev)
        for h in hs:
            api.bind_buffers(h, dX.data_ptr(), dY.data_ptr(), dC.data_ptr(), dR.data_ptr())
        api.set_factors(hs[0], X0, Y0)
        for h in hs:
            api.reset_stepsizes(h, params.stepsize)
        objs = []
        for _ in range(params.max_iter):
            for s, h in enumerate(hs):
                ml = row_bounds[s + 1] - row_bounds[s]
                if x_chunks > 1:
                    for j in range(x_chunks):
                        api.step_x_range(h, ml * j // x_chunks, ml * (j + 1) // x_chunks, params.min_stepsize)
                else:
                    api.step_x(h, params.min_stepsize)
            for h in hs:
                api.step_y(h, params.min_stepsize)
            objs.append(api.sum(hs[0], dC.data_ptr(), pa.n))
        X, Y = np.zeros_like(X0), np.zeros_like(Y0)
        api.get_factors(hs[0], X, Y)
        stats = [api.kernel_stats(h) for h in hs]
    finally:
        for h in hs:
            api.destroy(h)
    return np.array(objs), X, Y, stats
