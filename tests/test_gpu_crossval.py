"""-m gpu: glrm_hip_subset (device-side compaction of both Omega views) and the cross-validation drivers on the HIP engine
against the same drivers on the CPU oracle."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi, crossval

pytestmark = pytest.mark.gpu
TOL = 1e-5


def hip():
    return _capi.hip_api()


def model(rng, m=300, n=80, k=4, density=0.5, losses=None, rx=None, ry=None):
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) + 0.1 * rng.standard_normal((m, n))
    I, J = np.nonzero(rng.random((m, n)) < density)
    return L.GLRM(A, L.QuadLoss() if losses is None else losses, L.QuadReg(0.1) if rx is None else rx,
                  L.QuadReg(0.1) if ry is None else ry, k, obs=(I, J), X=rng.standard_normal((k, m)), Y=rng.standard_normal((k, n)))


@pytest.mark.parametrize("shape", [(300, 80, 0.5), (5000, 40, 0.9), (37, 11, 0.3), (20000, 64, 0.3)])
def test_subset_matches_oracle(shape):
    """Counts are exact; the child's objective and a short fit agree with the oracle's child built from the same tags."""
    m, n, dens = shape
    rng = np.random.default_rng(m + n)
    g = model(rng, m, n, 4, dens)
    rt = rng.integers(0, 4, len(g._colidx)).astype(np.uint8)
    ct = rng.integers(0, 4, len(g._rowidx)).astype(np.uint8)
    X, Y = np.asfortranarray(g.X), np.asfortranarray(g.Y)
    p = L.ProxGradParams(max_iter=6)
    res = {}
    for name, api in (("cpu", O.oracle_api()), ("hip", hip())):
        h = api.create(g.problem_arrays())
        out = []
        try:
            for match, inv in ((2, False), (2, True), (9, False), (9, True)):
                hc = api.subset(h, rt, ct, match, inv)
                try:
                    st = api.kernel_stats(hc)
                    Xc, Yc = X.copy(order="F"), Y.copy(order="F")
                    obj0 = api.objective(hc, Xc, Yc, True)
                    obj, _ = api.fit(hc, p, Xc, Yc)
                    out.append((st["nnz_rows"], st["nnz_cols"], obj0, obj, Xc, Yc))
                finally:
                    api.destroy(hc)
        finally:
            api.destroy(h)
        res[name] = out
    for c, g_ in zip(res["cpu"], res["hip"]):
        assert c[0] == g_[0] and c[1] == g_[1]
        assert g_[2] == pytest.approx(c[2], rel=1e-11)
        assert cases.rel_err(g_[3], c[3]) < TOL and cases.fro_err(g_[4], c[4]) < TOL and cases.fro_err(g_[5], c[5]) < TOL


def test_subset_child_outlives_parent_and_keeps_kernel_family():
    rng = np.random.default_rng(3)
    g = model(rng, 4000, 200, 8, 0.5)
    api = hip()
    h = api.create(g.problem_arrays(), tiled=2)
    tags_r = (rng.random(len(g._colidx)) < 0.2).astype(np.uint8)
    sp = crossval._Split(g)
    hc = api.subset(h, tags_r, tags_r[sp.perm], 1, True)
    api.destroy(h)
    try:
        assert api.kernel_stats(hc)["tiled"] == 3  # sorted lists stay sorted: the child runs the LDS-tiled sweeps too
        X, Y = np.asfortranarray(g.X), np.asfortranarray(g.Y)
        obj, _ = api.fit(hc, L.ProxGradParams(max_iter=5), X, Y)
        assert obj[-1] < obj[1]
    finally:
        api.destroy(hc)


def test_subset_rejects_dense_parent():
    rng = np.random.default_rng(4)
    A = rng.standard_normal((64, 48))
    g = L.GLRM(A, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), 16)
    api = hip()
    h = api.create(g.problem_arrays(dense=True))
    try:
        with pytest.raises(_capi.GLRMError) as ei:
            api.subset(h, np.zeros(1, np.uint8), np.zeros(1, np.uint8), 0, False)
        assert ei.value.code == _capi.ERR_UNSUPPORTED
    finally:
        api.destroy(h)


@pytest.mark.parametrize("kind", ["quad", "categorical"])
def test_cross_validate_hip_vs_oracle(kind):
    rng = np.random.default_rng(11)
    if kind == "quad":
        g = model(rng)
    else:
        kwargs, _ = cases.build_multidim_case("mnl")
        g = L.GLRM(**kwargs)
    tags = rng.integers(0, 3, int(g._rowptr[-1]))
    p = L.ProxGradParams(max_iter=12)
    out = {}
    for name, api in (("cpu", O.oracle_api()), ("hip", hip())):
        tr, te, trg, teg = L.cross_validate(g, nfolds=3, params=p, verbose=False, groups=tags, engine=api)
        out[name] = (tr, te, [t.X.copy() for t in trg])
        for t in trg + teg:
            t.close()
        g.close()
    assert cases.rel_err(out["hip"][0], out["cpu"][0]) < TOL and cases.rel_err(out["hip"][1], out["cpu"][1]) < TOL
    for a, b in zip(out["hip"][2], out["cpu"][2]):
        assert cases.fro_err(a, b) < TOL


def test_regularization_path_and_cv_by_iter_hip_vs_oracle():
    rng = np.random.default_rng(12)
    draws = None
    res = {}
    for name, api in (("cpu", O.oracle_api()), ("hip", hip())):
        g = model(np.random.default_rng(12))
        draws = np.random.default_rng(99).random(int(g._rowptr[-1]))
        tr, te, tt, rp = L.regularization_path(g, params=L.ProxGradParams(max_iter=10), reg_params=[5.0, 0.5, 0.05], holdout_proportion=0.2,
                                               verbose=False, groups=draws, engine=api)
        g2 = model(np.random.default_rng(12))
        ctr, cte = L.cv_by_iter(g2, 0.2, L.ProxGradParams(1.0, max_iter=6, abs_tol=0.0, rel_tol=0.0), verbose=False, groups=draws, engine=api)
        res[name] = (tr, te, ctr, cte)
    for a, b in zip(res["hip"], res["cpu"]):
        assert cases.rel_err(a, b) < TOL


def test_fused_and_unfused_drivers_agree_on_the_gpu():
    rng = np.random.default_rng(13)
    g = model(rng, 2000, 100, 6, 0.4)
    tags = rng.integers(0, 4, int(g._rowptr[-1]))
    p = L.HipProxGradParams(max_iter=8)
    a = L.cross_validate(g, nfolds=4, params=p, verbose=False, groups=tags, fused=True)
    b = L.cross_validate(g, nfolds=4, params=p, verbose=False, groups=tags, fused=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])  # the compacted views are the same arrays
