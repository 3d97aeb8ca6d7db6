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


GOLDEN_CASES = {"c1": (case_c1, 11), "nnmf": (case_nnmf, 12), "mixed": (case_mixed, 13), "kmeans": (case_kmeans, 14)}


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
