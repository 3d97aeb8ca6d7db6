"""SparseProxGradParams (src/algorithms/sparse_proxgrad.jl:22-134): the oracle against an independent numpy
transcription on a small QuadLoss model (CPU), and the HIP engine against the oracle (-m gpu)."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O


def numpy_sparse_proxgrad(A, I, J, X, Y, rx_scale, ry_scale, p):
    """Line-by-line numpy transcription for QuadLoss(1) + QuadReg on both factors (lists in obs order)."""
    m, n = A.shape
    k = X.shape[0]
    feats = [[] for _ in range(m)]
    exs = [[] for _ in range(n)]
    for i, j in zip(I, J):
        feats[i].append(j)
        exs[j].append(i)

    def objective(X, Y):
        err = 0.0
        for j in range(n):
            for i in exs[j]:
                err += (float(X[:, i] @ Y[:, j]) - A[i, j]) ** 2
        pen = 0.0
        for i in range(m):
            pen += rx_scale * float(np.sum(X[:, i] ** 2))
        for j in range(n):
            pen += ry_scale * float(np.sum(Y[:, j] ** 2))
        return err + pen

    bestX, bestY = X.copy(), Y.copy()
    X, Y = X.copy(), Y.copy()
    alpha = p.stepsize
    tol = p.abs_tol * len(I)
    ch = [objective(X, Y)]
    steps = 0
    for it in range(1, p.max_iter + 1):
        for _ in range(p.inner_iter):
            for e in range(m):
                g = np.zeros(k)
                for f in feats[e]:
                    g += 2 * (float(X[:, e] @ Y[:, f]) - A[e, f]) * Y[:, f]
                l = len(feats[e]) + 1
                X[:, e] = (X[:, e] + g * (-alpha / l)) / (1 + 2 * (alpha / l) * rx_scale)
        for _ in range(p.inner_iter):
            for f in range(n):
                g = np.zeros(k)
                for e in exs[f]:
                    g += 2 * (float(X[:, e] @ Y[:, f]) - A[e, f]) * X[:, e]
                l = len(exs[f]) + 1
                Y[:, f] = (Y[:, f] + g * (-alpha / l)) / (1 + 2 * (alpha / l) * ry_scale)
        obj = objective(X, Y)
        if obj < ch[-1]:
            ch.append(obj)
            bestX, bestY = X.copy(), Y.copy()
            alpha *= 1.05
            steps = max(1, steps + 1)
        else:
            alpha = alpha / max(1.5, -steps)
            X, Y = bestX.copy(), bestY.copy()
            steps = min(0, steps - 1)
        if (it > 10 and (steps > 3 and ch[-2] - obj < tol)) or alpha <= p.min_stepsize:
            break
    ch.append(ch[-1])
    return bestX, bestY, ch


def small_model(rng, m=25, n=18, k=3, density=0.6, rx=0.1, ry=0.2):
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) + 0.1 * rng.standard_normal((m, n))
    I, J = np.nonzero(rng.random((m, n)) < density)
    perm = rng.permutation(len(I))
    return A, I[perm], J[perm], rng.standard_normal((k, m)), rng.standard_normal((k, n)), rx, ry


@pytest.mark.parametrize("stepsize,inner", [(1.0, 1), (6.0, 1), (1.0, 2)])
def test_oracle_matches_numpy_transcription(stepsize, inner):
    rng = np.random.default_rng(17)
    A, I, J, X0, Y0, rx, ry = small_model(rng)
    p = L.SparseProxGradParams(stepsize, max_iter=40, inner_iter=inner)
    Xn, Yn, chn = numpy_sparse_proxgrad(A, I, J, X0, Y0, rx, ry, p)
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(rx), L.QuadReg(ry), 3, obs=(I, J), X=X0, Y=Y0)
    X, Y, ch = L.fit_b(g, p, verbose=False, engine=O.oracle_api())
    assert len(ch.objective) == len(chn)
    assert cases.rel_err(ch.objective, chn) < 1e-10
    assert cases.fro_err(X, Xn) < 1e-10 and cases.fro_err(Y, Yn) < 1e-10
    assert ch.objective[-1] == ch.objective[-2]  # the last value is recorded twice (:126-127)
    assert all(b < a for a, b in zip(ch.objective[:-2], ch.objective[1:-1]))  # only accepted iterations are recorded
    if stepsize == 6.0:  # a too-long first step is rejected and the step size shrinks: fewer records than iterations
        assert len(ch.objective) < 42


def test_defaults_and_repr(capsys):
    p = L.SparseProxGradParams()
    assert (p.stepsize, p.max_iter, p.inner_iter, p.abs_tol, p.min_stepsize) == (1.0, 100, 1, 1e-5, 0.01)
    assert repr(p) == "SparseProxGradParams(1.0, 100, 1, 1e-05, 0.01)"
    rng = np.random.default_rng(3)
    A, I, J, X0, Y0, rx, ry = small_model(rng)
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(rx), L.QuadReg(ry), 3, obs=(I, J), X=X0, Y=Y0)
    _, _, ch = L.fit_b(g, L.SparseProxGradParams(max_iter=3), engine=O.oracle_api())
    out = capsys.readouterr().out.splitlines()
    assert out[0].startswith("SparseProxGradParams(") and out[1] == "Fitting GLRM" and ch.name == "SparseProxGradGLRM"


# ------------------------------------------------------------------------------------------------ GPU

def _problem(rng, m, n, k, density, losses, rx, ry):
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) / np.sqrt(k) + 0.1 * rng.standard_normal((m, n))
    obs = None if density >= 1 else np.nonzero(rng.random((m, n)) < density)
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    g = L.GLRM(A, losses, rx, ry, k, obs=obs, X=X0, Y=Y0)
    return g, np.asfortranarray(X0), np.asfortranarray(Y0)


def _run_sparse(api, pa, X0, Y0, p, **kw):
    h = api.create(pa, **kw)
    try:
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        obj, _ = api.fit_sparse(h, p, X, Y)
    finally:
        api.destroy(h)
    return obj, X, Y


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["quad_k32", "nonneg_k8", "huber_l1_k5", "dense_k16", "long_cols"])
def test_hip_sparse_proxgrad_matches_oracle(case):
    from lowrankmodels.jl_amd import _capi
    rng = np.random.default_rng({"quad_k32": 1, "nonneg_k8": 2, "huber_l1_k5": 3, "dense_k16": 4, "long_cols": 5}[case])
    if case == "quad_k32":
        g, X0, Y0 = _problem(rng, 300, 120, 32, 0.3, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1))
    elif case == "nonneg_k8":
        g, X0, Y0 = _problem(rng, 200, 90, 8, 0.4, L.QuadLoss(), L.NonNegConstraint(), L.NonNegConstraint())
    elif case == "huber_l1_k5":
        g, X0, Y0 = _problem(rng, 150, 60, 5, 0.5, [L.HuberLoss() if f % 2 else L.QuadLoss(0.7) for f in range(60)], L.OneReg(0.05), L.QuadReg(0.2))
    elif case == "dense_k16":
        g, X0, Y0 = _problem(rng, 180, 130, 16, 1.0, L.QuadLoss(), L.QuadReg(0.1), L.ZeroReg())
    else:
        g, X0, Y0 = _problem(rng, 5000, 10, 16, 0.7, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1))
    p = L.SparseProxGradParams(2.0, max_iter=30)
    O.set_threads(4)
    o_c, X_c, Y_c = _run_sparse(O.oracle_api(), g.problem_arrays(), X0, Y0, p)
    variants = [dict(tiled=1), dict(tiled=2)]
    for kw in variants:
        o_g, X_g, Y_g = _run_sparse(_capi.hip_api(), g.problem_arrays(), X0, Y0, p, **kw)
        assert len(o_g) == len(o_c), (kw, len(o_g), len(o_c))
        assert cases.rel_err(o_g, o_c) < 1e-5 and cases.fro_err(X_g, X_c) < 1e-5 and cases.fro_err(Y_g, Y_c) < 1e-5, kw
    if case == "dense_k16":  # the same model through the dense MFMA hand-over
        o_g, X_g, Y_g = _run_sparse(_capi.hip_api(), g.problem_arrays(dense=True), X0, Y0, p)
        assert len(o_g) == len(o_c) and cases.rel_err(o_g, o_c) < 1e-5 and cases.fro_err(X_g, X_c) < 1e-5
        # host level
        X, Y, ch = L.fit_b(g, L.SparseProxGradParams(2.0, max_iter=30), verbose=False)
        assert cases.rel_err(ch.objective, o_c) < 1e-5
        g.close()


def test_default_solver_follows_the_reference_dispatch():
    """fit!(glrm) without params: SparseMatrixCSC input -> SparseProxGradParams, dense input -> ProxGradParams (src/fit.jl:13-18)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    A = sp.random(30, 20, density=0.3, random_state=1, format="csc")
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), 3, rng=rng)
    _, _, ch = L.fit_b(g, verbose=False, engine=O.oracle_api())
    assert ch.name == "SparseProxGradGLRM"
    gd = L.GLRM(A.toarray(), L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), 3, rng=rng)
    _, _, chd = L.fit_b(gd, verbose=False, engine=O.oracle_api())
    assert chd.name == "ProxGradGLRM"
