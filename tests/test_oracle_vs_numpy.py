"""Second, independently written restatement of fit!(::GLRM, ::ProxGradParams) (src/algorithms/proxgrad.jl:34-220): a
dense numpy transcription that allocates XY = X'Y and evaluates full rows / columns exactly like the reference does,
generic over the Python loss / regularizer mirrors.  The C oracle (observed-only arithmetic, different language, different
data layout) must agree with it to rounding on every recorded objective, on X and Y, and on the line-search state --
SURVEY.md section 8(c) item (4)."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O


def numpy_proxgrad(A, losses, rx, ry, feats, exs, X, Y, p):
    m, n = A.shape
    X, Y = X.copy(), Y.copy()
    XY = X.T @ Y                                            # gemm!('T','N',1.0,X,Y,0.0,XY)  :66
    alpharow = p.stepsize * np.ones(m)                      # :69-70
    alphacol = p.stepsize * np.ones(n)
    scaled_abs_tol = p.abs_tol * sum(len(f) for f in feats)  # :72

    def row_objective(i, x):                                # src/evaluate_fit.jl:24-38
        xy = x @ Y
        err = 0.0
        for j in feats[i]:
            err += losses[j].evaluate(xy[j], A[i, j])
        return err + rx[i].evaluate(x)

    def col_objective(j, y):                                # src/evaluate_fit.jl:39-55
        xy = X.T @ y
        err = 0.0
        for i in exs[j]:
            err += losses[j].evaluate(xy[i], A[i, j])
        return err + ry[j].evaluate(y)

    obj0 = 0.0                                              # objective(glrm, X, Y, XY)  src/evaluate_fit.jl:4-23
    for j in range(n):
        for i in exs[j]:
            obj0 += losses[j].evaluate(XY[i, j], A[i, j])
    obj0 += sum(rx[i].evaluate(X[:, i]) for i in range(m)) + sum(ry[j].evaluate(Y[:, j]) for j in range(n))
    ch = [obj0]
    obj_by_col = np.zeros(n)
    for it in range(1, p.max_iter + 1):
        if p.inner_iter_X > 1 or p.inner_iter_Y > 1:        # :112-115
            alpharow[:] = p.stepsize
            alphacol[:] = p.stepsize
        for _ in range(p.inner_iter_X):
            for e in range(m):                              # :118-156
                g = np.zeros(X.shape[0])
                for f in feats[e]:
                    g += losses[f].grad(XY[e, f], A[e, f]) * Y[:, f]
                l = len(feats[e]) + 1
                obj_old = row_objective(e, X[:, e])
                while alpharow[e] > p.min_stepsize:
                    stepsize = alpharow[e] / l
                    newx = np.asarray(rx[e].prox(X[:, e] - stepsize * g, stepsize), dtype=float)
                    if row_objective(e, newx) < obj_old:
                        X[:, e] = newx
                        alpharow[e] *= 1.05
                        break
                    alpharow[e] *= .7
                    if alpharow[e] < p.min_stepsize:
                        alpharow[e] = p.min_stepsize * 1.1
                        break
            XY = X.T @ Y                                    # :157
        for _ in range(p.inner_iter_Y):
            for f in range(n):                              # :162-201
                G = np.zeros(Y.shape[0])
                for e in exs[f]:
                    G += losses[f].grad(XY[e, f], A[e, f]) * X[:, e]
                l = len(exs[f]) + 1
                obj_by_col[f] = col_objective(f, Y[:, f])
                while alphacol[f] > p.min_stepsize:
                    stepsize = alphacol[f] / l
                    newy = np.asarray(ry[f].prox(Y[:, f] - stepsize * G, stepsize), dtype=float)
                    new_obj = col_objective(f, newy)
                    if new_obj < obj_by_col[f]:
                        Y[:, f] = newy
                        alphacol[f] *= 1.05
                        obj_by_col[f] = new_obj
                        break
                    alphacol[f] *= .7
                    if alphacol[f] < p.min_stepsize:
                        alphacol[f] = p.min_stepsize * 1.1
                        break
            XY = X.T @ Y                                    # :202
        obj = float(np.sum(obj_by_col))                     # :205
        ch.append(obj)
        dec = ch[-2] - obj
        if it > 10 and (dec < scaled_abs_tol or dec / obj < p.rel_tol):  # :210-213
            break
    return X, Y, ch, alpharow, alphacol


def model(name, rng):
    if name == "quad_quadreg":
        m, n, k = 30, 22, 3
        A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) + 0.1 * rng.standard_normal((m, n))
        losses, rx, ry = [L.QuadLoss() for _ in range(n)], [L.QuadReg(0.1)] * m, [L.QuadReg(0.2)] * n
        p = L.ProxGradParams(max_iter=25)
    elif name == "nnmf_inf_start":
        m, n, k = 24, 18, 3
        A = rng.random((m, k)) @ rng.random((k, n))
        losses, rx, ry = [L.QuadLoss() for _ in range(n)], [L.NonNegConstraint()] * m, [L.NonNegConstraint()] * n
        p = L.ProxGradParams(max_iter=25)
    elif name == "mixed_losses_per_row_regs":
        m, n, k = 26, 15, 2
        Z = rng.standard_normal((m, k)) @ rng.standard_normal((k, n))
        A = np.zeros((m, n))
        losses = []
        for f in range(n):
            kind = f % 5
            if kind == 0:
                A[:, f] = Z[:, f]; losses.append(L.QuadLoss(0.8))
            elif kind == 1:
                A[:, f] = rng.random(m) < 0.5; losses.append(L.LogisticLoss(1.2))
            elif kind == 2:
                A[:, f] = np.clip(np.round(3 + Z[:, f]), 1, 5); losses.append(L.OrdinalHingeLoss(1, 5))
            elif kind == 3:
                A[:, f] = Z[:, f]; losses.append(L.HuberLoss(1.0, crossover=0.7))
            else:
                A[:, f] = rng.random(m) < 0.4; losses.append(L.WeightedHingeLoss(1.0, case_weight_ratio=1.5))
        kinds = [L.QuadReg(0.3), L.OneReg(0.1), L.NonNegConstraint(), L.ZeroReg()]
        rx, ry = [kinds[i % 4] for i in range(m)], [L.QuadReg(0.2)] * n
        p = L.ProxGradParams(max_iter=15)
    else:  # kmeans_inner10
        m, n, k = 20, 12, 3
        A = (rng.standard_normal((k, n)) * 3)[rng.integers(0, k, m)] + 0.1 * rng.standard_normal((m, n))
        losses, rx, ry = [L.QuadLoss() for _ in range(n)], [L.UnitOneSparseConstraint()] * m, [L.ZeroReg()] * n
        p = L.ProxGradParams(max_iter=12, inner_iter=4)
    mask = rng.random((m, n)) < 0.7
    feats = [list(np.flatnonzero(mask[i])) for i in range(m)]
    exs = [list(np.flatnonzero(mask[:, j])) for j in range(n)]
    return A, losses, rx, ry, feats, exs, rng.standard_normal((k, m)), rng.standard_normal((k, n)), p, k


@pytest.mark.parametrize("name", ["quad_quadreg", "nnmf_inf_start", "mixed_losses_per_row_regs", "kmeans_inner10"])
def test_c_oracle_agrees_with_dense_numpy_transcription(name):
    rng = np.random.default_rng({"quad_quadreg": 1, "nnmf_inf_start": 2, "mixed_losses_per_row_regs": 3, "kmeans_inner10": 4}[name])
    A, losses, rx, ry, feats, exs, X0, Y0, p, k = model(name, rng)
    Xn, Yn, chn, ar, ac = numpy_proxgrad(A, losses, rx, ry, feats, exs, X0, Y0, p)
    g = L.GLRM(A, losses, rx, ry, k, observed_features=feats, observed_examples=exs, X=X0, Y=Y0)
    api, lib = O.oracle_api(), O.oracle_lib()
    O.set_threads(1)
    h = api.create(g.problem_arrays())
    try:
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        obj, _ = api.fit(h, p, X, Y)
        a_r, a_c = np.zeros(len(feats)), np.zeros(len(exs))
        lib.glrm_cpu_get_stepsizes(h, a_r.ctypes.data, a_c.ctypes.data)
    finally:
        api.destroy(h)
    assert len(obj) == len(chn)
    assert cases.rel_err(obj, chn) < 1e-9
    assert cases.fro_err(X, Xn) < 1e-9 and cases.fro_err(Y, Yn) < 1e-9
    # identical line-search decisions: the per-row / per-column step sizes coincide
    np.testing.assert_allclose(a_r, ar, rtol=1e-12)
    np.testing.assert_allclose(a_c, ac, rtol=1e-12)


# ------------------------------------------------------------------------------------------------------------------
# Multi-dimensional losses, block regularizers, offsets: the same transcription with Y's column spans (get_yidxs,
# src/losses.jl:76-93), vector gradients (gemm! branches of proxgrad.jl:126-131,169-174) and k x d prox blocks.

def numpy_proxgrad_general(A, losses, rx, ry, feats, exs, X, Y, p):
    m, n = A.shape
    k = X.shape[0]
    X, Y = X.copy(), Y.copy()
    yidxs = L.get_yidxs(losses)

    def sl(j):
        return slice(*yidxs[j])

    def lev(j, u, a):                                       # scalar columns see a Float64, vector ones a Vector
        return losses[j].evaluate(u[0], a) if losses[j].embedding_dim == 1 else losses[j].evaluate(u, a)

    def lgrad(j, u, a):
        return np.atleast_1d(losses[j].grad(u[0], a)) if losses[j].embedding_dim == 1 else np.asarray(losses[j].grad(u, a))

    def yblock(j, Yb):                                      # what the reference hands to evaluate / prox: vf[f]
        return Yb[:, 0] if losses[j].embedding_dim == 1 else Yb

    XY = X.T @ Y
    alpharow, alphacol = p.stepsize * np.ones(m), p.stepsize * np.ones(n)
    scaled_abs_tol = p.abs_tol * sum(len(f) for f in feats)

    def row_objective(i, x):
        xy = x @ Y
        err = 0.0
        for j in feats[i]:
            err += lev(j, xy[sl(j)], A[i, j])
        return err + rx[i].evaluate(x)

    def col_objective(j, yb):
        xy = X.T @ yb.reshape(k, -1)
        err = 0.0
        for i in exs[j]:
            err += lev(j, xy[i], A[i, j])
        return err + ry[j].evaluate(yblock(j, yb.reshape(k, -1)))

    obj0 = 0.0
    for j in range(n):
        for i in exs[j]:
            obj0 += lev(j, XY[i, sl(j)], A[i, j])
    obj0 += sum(rx[i].evaluate(X[:, i]) for i in range(m)) + sum(ry[j].evaluate(yblock(j, Y[:, sl(j)])) for j in range(n))
    ch = [obj0]
    obj_by_col = np.zeros(n)
    for it in range(1, p.max_iter + 1):
        if p.inner_iter_X > 1 or p.inner_iter_Y > 1:
            alpharow[:] = p.stepsize
            alphacol[:] = p.stepsize
        for _ in range(p.inner_iter_X):
            for e in range(m):
                g = np.zeros(k)
                for f in feats[e]:
                    g += Y[:, sl(f)] @ lgrad(f, XY[e, sl(f)], A[e, f])
                l = len(feats[e]) + 1
                obj_old = row_objective(e, X[:, e])
                while alpharow[e] > p.min_stepsize:
                    stepsize = alpharow[e] / l
                    newx = np.asarray(rx[e].prox(X[:, e] - stepsize * g, stepsize), dtype=float)
                    if row_objective(e, newx) < obj_old:
                        X[:, e] = newx
                        alpharow[e] *= 1.05
                        break
                    alpharow[e] *= .7
                    if alpharow[e] < p.min_stepsize:
                        alpharow[e] = p.min_stepsize * 1.1
                        break
            XY = X.T @ Y
        for _ in range(p.inner_iter_Y):
            for f in range(n):
                d = losses[f].embedding_dim
                G = np.zeros((k, d))
                for e in exs[f]:
                    G += np.outer(X[:, e], lgrad(f, XY[e, sl(f)], A[e, f]))
                l = len(exs[f]) + 1
                obj_by_col[f] = col_objective(f, Y[:, sl(f)])
                while alphacol[f] > p.min_stepsize:
                    stepsize = alphacol[f] / l
                    newy = np.asarray(ry[f].prox(yblock(f, Y[:, sl(f)] - stepsize * G), stepsize), dtype=float).reshape(k, d)
                    new_obj = col_objective(f, newy)
                    if new_obj < obj_by_col[f]:
                        Y[:, sl(f)] = newy
                        alphacol[f] *= 1.05
                        obj_by_col[f] = new_obj
                        break
                    alphacol[f] *= .7
                    if alphacol[f] < p.min_stepsize:
                        alphacol[f] = p.min_stepsize * 1.1
                        break
            XY = X.T @ Y
        obj = float(np.sum(obj_by_col))
        ch.append(obj)
        dec = ch[-2] - obj
        if it > 10 and (dec < scaled_abs_tol or dec / obj < p.rel_tol):
            break
    return X, Y, ch, alpharow, alphacol


def expand(kwargs):
    """GLRM constructor arguments -> per-column / per-row Python objects and the two observation views."""
    g = L.GLRM(**kwargs)
    feats = [list(g._colidx[g._rowptr[i]:g._rowptr[i + 1]]) for i in range(g.m)]
    exs = [list(g._rowidx[g._colptr[j]:g._colptr[j + 1]]) for j in range(g.n)]
    return g, np.asarray(kwargs["A"], dtype=float), g.losses, g.rx, g.ry, feats, exs


@pytest.mark.parametrize("name", list(cases.MULTIDIM_CASES))
def test_c_oracle_multidim_agrees_with_numpy_transcription(name):
    kwargs, p = cases.build_multidim_case(name)
    if name in ("mnl", "mnl_ordinal", "loss_test"):
        p = L.ProxGradParams(max_iter=8)  # the pure-Python loops are slow; the early iterations carry the line search
    g, A, losses, rx, ry, feats, exs = expand(kwargs)
    X0, Y0 = kwargs["X"], kwargs["Y"]
    Xn, Yn, chn, ar, ac = numpy_proxgrad_general(A, losses, rx, ry, feats, exs, X0, Y0, p)
    api, lib = O.oracle_api(), O.oracle_lib()
    O.set_threads(1)
    h = api.create(g.problem_arrays())
    try:
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        obj, _ = api.fit(h, p, X, Y)
        a_r, a_c = np.zeros(len(feats)), np.zeros(len(exs))
        lib.glrm_cpu_get_stepsizes(h, a_r.ctypes.data, a_c.ctypes.data)
    finally:
        api.destroy(h)
    assert len(obj) == len(chn)
    assert cases.rel_err(obj, chn) < 1e-9
    assert cases.fro_err(X, Xn) < 1e-9 and cases.fro_err(Y, Yn) < 1e-9
    np.testing.assert_allclose(a_r, ar, rtol=1e-12)
    np.testing.assert_allclose(a_c, ac, rtol=1e-12)
