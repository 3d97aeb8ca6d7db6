"""init_svd! (src/initialize.jl:35-132): the oracle's dense restatement against a numpy transcription (np.linalg.svd in place of
Arpack's svds), on scalar, categorical and ordinal models with missing entries."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O


def numpy_init_svd(g):
    """Transcription of init_svd! for the losses of include/glrm_hip.h -> (X, Y, singular values)."""
    m, n, k = g.m, g.n, g.k
    yidxs = L.get_yidxs(g.losses)
    d = yidxs[-1][1]
    A = np.asarray(g.A, dtype=float)
    exs = [g._rowidx[g._colptr[f]:g._colptr[f + 1]] for f in range(n)]
    Areal = np.zeros((m, d))
    for f, (lo, (y0, y1)) in enumerate(zip(g.losses, yidxs)):      # :47-80
        e = exs[f]
        if y1 - y0 == 1:
            Areal[e, y0] = A[e, f]
        elif isinstance(lo, (L.MultinomialLoss, L.OvALoss)):        # CategoricalDomain: levels 1:max
            for il in range(lo.max):
                Areal[e, y0 + il] = np.where(A[e, f] == il + 1, 1, -1)
        else:                                                        # OrdinalDomain: one column per level but the last
            for il in range(lo.max - 1):
                Areal[e, y0 + il] = np.where(A[e, f] > il + 1, 1, -1)
    means, stds, Astd = np.zeros(d), np.zeros(d), np.zeros((m, d))
    for f, (y0, y1) in enumerate(yidxs):                             # :83-98
        for j in range(y0, y1):
            nomissing = Areal[exs[f], j]
            with np.errstate(all="ignore"):
                means[j] = np.mean(nomissing) if len(nomissing) else np.nan
                stds[j] = np.std(nomissing, ddof=1) if len(nomissing) > 1 else np.nan
            if np.isnan(means[j]):
                means[j] = 1
            if stds[j] < 1e-10 or np.isnan(stds[j]):
                stds[j] = 1
            Astd[exs[f], j] = Areal[exs[f], j] - means[j]
    Astd *= m * n / int(g._rowptr[-1])                               # :113
    U, S, Vt = np.linalg.svd(Astd, full_matrices=False)              # svds(Astd, nsv = k) :121
    X = np.diag(np.sqrt(S[:k])) @ U[:, :k].T                         # :129
    Y = np.diag(np.sqrt(S[:k])) @ Vt[:k] @ np.diag(stds)             # :130
    return X, Y, S[:k]


def scalar_model(rng, m=60, n=35, k=4):
    A = rng.standard_normal((m, 3)) @ rng.standard_normal((3, n)) * 2 + rng.standard_normal((m, n)) * 0.3 + rng.standard_normal(n) * 3
    A[:, 5] = rng.random(m) < 0.4
    losses = [L.LogisticLoss() if f == 5 else (L.HuberLoss() if f % 7 == 0 else L.QuadLoss()) for f in range(n)]
    I, J = np.nonzero(rng.random((m, n)) < 0.7)
    return L.GLRM(A, losses, L.QuadReg(0.1), L.QuadReg(0.1), k, obs=(I, J))


def model_of(name):
    rng = np.random.default_rng(7)
    if name == "scalar":
        return scalar_model(rng)
    if name == "constant_and_empty_columns":
        g = scalar_model(rng)
        A = np.array(g.A); A[:, 3] = 2.5
        I = np.repeat(np.arange(g.m), np.diff(g._rowptr)); J = g._colidx.astype(np.int64)
        keep = (J != 9) & ~((J == 11) & (I > 0))                    # column 9 unobserved, column 11 observed once
        return L.GLRM(A, g.losses, L.QuadReg(0.1), L.QuadReg(0.1), g.k, obs=(I[keep], J[keep]))
    kwargs, _ = cases.build_multidim_case(name)
    kwargs = dict(kwargs)
    if "observed_features" in kwargs:                                # same entries in both views
        feats = kwargs.pop("observed_features"); kwargs.pop("observed_examples")
        I = np.repeat(np.arange(len(feats)), [len(f) for f in feats]); J = np.concatenate([np.asarray(f, dtype=np.int64) for f in feats])
        kwargs["obs"] = (I, J)
    return L.GLRM(**kwargs)


@pytest.mark.parametrize("name", ["scalar", "constant_and_empty_columns", "categorical_mix", "ordinal_offsets", "loss_test", "mnl"])
def test_oracle_init_svd_matches_numpy_transcription(name):
    g = model_of(name)
    Xn, Yn, Sn = numpy_init_svd(g)
    L.init_svd_(g, engine=O.oracle_api())
    info = g._init_svd_info
    np.testing.assert_allclose(info["singular_values"], Sn, rtol=1e-10)
    # each component is determined up to its sign; the product is unique (distinct singular values)
    np.testing.assert_allclose(g.X.T @ g.Y, Xn.T @ Yn, rtol=0, atol=1e-9 * np.abs(Xn.T @ Yn).max())
    np.testing.assert_allclose(np.abs(g.X), np.abs(Xn), rtol=0, atol=1e-8 * np.abs(Xn).max())
    np.testing.assert_allclose(np.abs(g.Y), np.abs(Yn), rtol=0, atol=1e-8 * np.abs(Yn).max())


def test_init_svd_improves_the_start_and_is_a_fit_warm_start():
    """notebook :308-317 vs :587: the SVD start is far closer than randn."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((80, 3)) @ rng.standard_normal((3, 40)) + 0.05 * rng.standard_normal((80, 40))
    A -= A.mean(axis=0)            # the reference's start reproduces the CENTRED data (its offset branch never runs) ...
    A /= A.std(axis=0, ddof=1)     # ... times diag(stds), although Astd was never divided by them (:106-108 vs :130)
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.01), L.QuadReg(0.01), 3)
    api = O.oracle_api()
    before = L.objective(g, engine=api)
    L.init_svd_(g, engine=api)
    after = L.objective(g, engine=api)
    assert after < 0.01 * before
    X, Y, ch = L.fit_b(g, L.ProxGradParams(max_iter=10), verbose=False, engine=api)
    assert ch.objective[0] == pytest.approx(after, rel=1e-12) and ch.objective[-1] <= ch.objective[0]


def test_argument_checks():
    kwargs, _ = cases.build_golden_case("mixed")  # duplicates / views that disagree
    g = L.GLRM(**kwargs)
    with pytest.raises(ValueError):
        L.init_svd_(g, engine=O.oracle_api())
    rng = np.random.default_rng(5)
    g2 = L.GLRM(rng.standard_normal((4, 30)), L.QuadLoss(), L.QuadReg(), L.QuadReg(), 5)
    with pytest.raises(L.GLRMError):
        L.init_svd_(g2, engine=O.oracle_api())  # k > m
