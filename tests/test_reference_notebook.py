"""The one trajectory of `fit!(glrm, ProxGradParams())` the reference holds, as a golden vector.

`examples/LowRankModelsDemo-v1.1.0.ipynb` (Julia 1.1.0) runs, after `Random.seed!(1)`:

    m,n = 20,10;  A = rand(m,2)*rand(2,n)                       # cell at :270-283
    Ω = [(rand(1:m), rand(1:n)) for iobs in 1:100]              # 100 entries WITH replacement: duplicates stay in the lists
    glrm = GLRM(A, QuadLoss(), NonNegConstraint(), NonNegConstraint(), 5, obs=Ω)   # X = randn(5,20), then Y = randn(5,10) (src/glrm.jl:31)
    X,Y,ch = fit!(glrm)                                         # default ProxGradParams (src/algorithms/proxgrad.jl:12-28)

and its saved output holds the objective at iterations 10, 20, ..., 100 in full precision (:308-317) and the corner entries of the fitted
X and Y at six digits (:518-523, :543-548).  `tests/julia_rng.py` restates Julia 1.1's MersenneTwister (dSFMT-19937 + rand / rand(a:b)
/ randn), so the cell's random inputs are rebuilt here bit for bit and the printed numbers become known answers for the whole path:
observation lists with duplicates, N(0,1) start outside the constraint set (objective[1] = Inf), 100 outer iterations of backtracking
line search + NonNeg prox, run into max_iter.

Measured: the oracle reproduces iteration 10 bit for bit and all ten printed objectives to <= 6e-12 relative (summation order inside a
row / column differs from Julia's BLAS); the HIP engine agrees with the notebook to the same level.

Two more cells of the same notebook print trajectories and are pinned the same way:
  * `init_svd!(glrm); X,Y,ch_svd = fit!(glrm)` on the model above (:568-595): objective at iterations 10 / 20 / 30 in full precision,
    the first ten and last ten entries of ch_svd.objective at six digits, 32 entries in all.  Arpack's choice of sign for each singular
    pair is not reproducible (and matters under NonNegConstraint), so the test looks for THE sign assignment among the 2^5 whose
    first step matches and requires the whole trajectory of that one to match.
  * `Random.seed!(1); A_sparse = sprandn(10, 10, .8)`, `GLRM(A_sparse, HuberLoss(), QuadReg(.1), QuadReg(.1), 5)`, `init_svd!`, `fit!`
    (:870-1030): a sparse A makes SparseProxGradParams the default solver (src/algorithms/sparse_proxgrad.jl).  sprandn is restated
    (sprand_IJ + randsubseq of SparseArrays 1.1) and checked against the printed entries; the objective is invariant under the sign of
    a singular pair here (QuadReg on both factors), so no search is needed.
"""
import numpy as np
import pytest

import julia_rng as J
import lowrankmodels.jl_amd as L
import oracle as O

# examples/LowRankModelsDemo-v1.1.0.ipynb:308-317
NOTEBOOK_OBJECTIVE = {
    10: 30.96601471516671, 20: 25.906874109924694, 30: 21.032954794013904, 40: 18.652747843346216, 50: 17.555095496418712,
    60: 17.040150781258383, 70: 16.676535553975437, 80: 16.36043529444562, 90: 16.04190265061905, 100: 15.739454018044162,
}
# :518-523 -- X (5 x 20): columns 1-4 and 17-20 as printed
NOTEBOOK_X = {
    0: [0.0869609, 0.0, 0.0, 0.0, 0.0], 1: [0.133926, 0.0853408, 0.0434652, 0.136009, 0.00343981], 2: [0.12529, 0.125362, 0.0, 0.0, 0.0],
    3: [0.0] * 5, 16: [0.0] * 5, 17: [0.0] * 5, 18: [0.113651, 0.00705357, 0.0267825, 0.0180746, 0.0513907],
    19: [0.395263, 0.0, 0.00186851, 0.0, 2.8092e-5],
}
# :543-548 -- Y (5 x 10): columns 1-3 and 8-10 as printed
NOTEBOOK_Y = {
    0: [2.11424, 2.93206, 0.0629576, 0.185631, 0.130496], 1: [2.0145, 2.26443, 0.00173739, 0.210268, 0.21822],
    2: [0.689844, 0.623894, 0.00235652, 0.000980211, 0.000739932], 7: [1.03779, 1.19131, 2.90583, 0.0, 0.232767],
    8: [0.491886, 3.23841, 2.24857, 1.87045, 2.49313], 9: [0.657266, 1.88255, 6.0587, 2.61232, 0.19638],
}


def test_julia_rng_known_values():
    """Known outputs of Julia (<= 1.6) for the default MersenneTwister."""
    r = J.MersenneTwister(1)
    assert [r.rand() for _ in range(3)] == [0.23603334566204692, 0.34651701419196046, 0.3127069683360675]
    assert J.MersenneTwister(0).rand() == 0.8236475079774124
    assert J.MersenneTwister(1234).rand() == 0.5908446386657102
    assert J.MersenneTwister(1).randn() == 0.2972879845354616
    # remembered entries of normal.jl's tables (1-based there)
    assert J.KI[0] == 0x7799ec012f7b2 and J.KI[1] == 0 and J.KI[2] == 0x6045f4c7de363
    assert J.WI[0] == 1.7367254121602630e-15 and J.WI[1] == 9.5586603514556339e-17 and J.FI[0] == 1.0 and J.FI[1] == 9.7710170126767082e-01
    # the cache refill (1002 doubles per gen_rand) and rand(a:b)'s rejection loop keep the stream intact
    r = J.MersenneTwister(1)
    draws = [r.rand_range(1, 20) for _ in range(3000)]
    assert min(draws) == 1 and max(draws) == 20 and abs(np.mean(draws) - 10.5) < 0.3
    z = np.array([r.randn() for _ in range(20000)])
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03 and abs((np.abs(z) > 3.6541528853610088).mean() - 2.6e-4) < 4e-4


def notebook_model():
    """The cell's random inputs, in the order Julia draws them."""
    r = J.MersenneTwister(1)
    m, n, k = 20, 10, 5
    A = r.rand_matrix(m, 2) @ r.rand_matrix(2, n)
    obs = [(r.rand_range(1, m) - 1, r.rand_range(1, n) - 1) for _ in range(100)]  # tuple fields are drawn left to right
    X0 = r.randn_matrix(k, m)
    Y0 = r.randn_matrix(k, n)
    assert len(set(obs)) < len(obs)  # sampled with replacement: the lists carry duplicates (sort_observations, src/glrm.jl keeps them)
    return L.GLRM(A, L.QuadLoss(), L.NonNegConstraint(), L.NonNegConstraint(), k, obs=obs, X=X0, Y=Y0)


def check_against_the_notebook(X, Y, ch, rtol):
    obj = np.array(ch.objective)
    assert len(obj) == 101 and obj[0] == np.inf  # "first entry is infinite, since initial factors are not nonnegative" (:330)
    for it, want in NOTEBOOK_OBJECTIVE.items():
        assert abs(obj[it] - want) <= rtol * want, (it, obj[it], want)
    for shown, got in ((NOTEBOOK_X, X), (NOTEBOOK_Y, Y)):
        for col, vals in shown.items():
            for i, v in enumerate(vals):
                # six significant digits as printed; entries printed as 0.0 are exact zeros of the NonNeg prox
                assert (got[i, col] == 0.0) if v == 0.0 else abs(got[i, col] - v) <= 6e-6 * abs(v), (col, i, got[i, col], v)


def test_oracle_reproduces_the_notebook_trajectory():
    O.set_threads(1)
    g = notebook_model()
    X, Y, ch = L.fit_b(g, L.ProxGradParams(), verbose=False, engine=O.oracle_api())
    check_against_the_notebook(X, Y, ch, rtol=1e-10)
    assert np.array(ch.objective)[10] == NOTEBOOK_OBJECTIVE[10]  # ten iterations in: the same double
    g.close()


def test_oracle_threads_do_not_change_the_notebook_trajectory():
    O.set_threads(4)
    g = notebook_model()
    X, Y, ch = L.fit_b(g, L.ProxGradParams(), verbose=False, engine=O.oracle_api())
    check_against_the_notebook(X, Y, ch, rtol=1e-10)
    g.close()


def test_dense_numpy_transcription_reproduces_the_notebook_trajectory_to_the_last_digits():
    """The line-by-line numpy transcription of proxgrad.jl (dense XY, gemm-shaped products like the reference's BLAS calls), which the
    oracle is checked against in test_oracle_vs_numpy.py, lands on the printed doubles: 7 of 10 identical, the rest one ulp away."""
    import test_oracle_vs_numpy as N
    g = notebook_model()
    feats, exs = [list(f) for f in g.observed_features], [list(e) for e in g.observed_examples]
    Xn, Yn, chn, _, _ = N.numpy_proxgrad(g.A, g.losses, g.rx, g.ry, feats, exs, np.array(g.X), np.array(g.Y), L.ProxGradParams())

    class CH:
        objective = chn
    check_against_the_notebook(Xn, Yn, CH, rtol=1e-13)  # measured here: <= 2e-16 (the host BLAS decides the last bit)
    g.close()


# ---------------------------------------------------------------------------------------------------------------------------
# init_svd! cell (:568-595)
NOTEBOOK_SVD_OBJECTIVE = {10: 0.13447604673539088, 20: 0.038027469752441506, 30: 0.01086888732733188}
NOTEBOOK_SVD_HEAD = [np.inf, 28.4395, 9.14371, 0.864888, 0.37692, 0.272977, 0.221574, 0.193896, 0.17122, 0.151837]
NOTEBOOK_SVD_TAIL = [0.0289067, 0.0249007, 0.0217853, 0.0190616, 0.0170927, 0.0154683, 0.0137965, 0.0120712, 0.0108689, 0.00989485]


NOTEBOOK_IMPUTE = [  # :815-835 -- impute(glrm) after the init_svd! fit: columns 1-4 and 8-10 of the 20 x 10 matrix as printed
    [0.201064, 0.178928, 0.0936361, 0.338963, 0.0610706, 0.0345337, 0.255634],
    [0.626425, 0.498201, 0.404066, 0.52571, 0.35284, 0.163703, 0.485699],
    [0.889128, 0.565398, 0.590757, 0.727527, 0.501572, 0.273879, 0.524708],
    [0.264933, 0.213167, 0.172958, 0.30727, 0.140076, 0.0812656, 0.20812],
    [0.849472, 0.627437, 0.40914, 0.780748, 0.346765, 0.113373, 0.797801],
    [0.488824, 0.434675, 0.350332, 0.670804, 0.271115, 0.174788, 0.472898],
    [0.620052, 0.641855, 0.322417, 0.565983, 0.29541, 0.0685478, 0.72269],
    [1.1255, 1.02346, 0.63119, 1.2256, 0.527708, 0.210468, 1.20615],
    [0.214933, 0.198515, 0.102426, 0.335253, 0.0716396, 0.0341962, 0.2766],
    [0.79803, 0.865126, 0.424412, 0.781505, 0.385092, 0.0947092, 0.968798],
    [0.612195, 0.538832, 0.363826, 0.736084, 0.292544, 0.14433, 0.624619],
    [0.325503, 0.351202, 0.185431, 0.469194, 0.148028, 0.0699904, 0.372757],
    [0.371371, 0.326883, 0.187448, 0.429956, 0.152637, 0.0570415, 0.403141],
    [1.15949, 0.916536, 0.710041, 1.34629, 0.565187, 0.305434, 1.06717],
    [0.223678, 0.208483, 0.0833057, 0.284272, 0.0673465, 0.012477, 0.251368],
    [0.556494, 0.35883, 0.295561, 0.545181, 0.239097, 0.112382, 0.451415],
    [0.504458, 0.394671, 0.306674, 0.526402, 0.252477, 0.126227, 0.440751],
    [0.241705, 0.312739, 0.140464, 0.186293, 0.140516, 0.0255899, 0.301894],
    [0.253238, 0.271053, 0.157952, 0.36181, 0.124748, 0.0631709, 0.308106],
    [0.809283, 0.775808, 0.489043, 1.01611, 0.395559, 0.19164, 0.872817],
]


def init_svd_numpy(A, exs, k, nobs):
    """src/initialize.jl:83-131 for scalar losses without offset: centre the observed entries of each column (means and stds over the
    LIST, duplicates counted), zero elsewhere, scale by m n / |obs|, top-k SVD, X = sqrt(S) U', Y = sqrt(S) V' diag(stds)."""
    m, n = A.shape
    Astd, stds = np.zeros((m, n)), np.ones(n)
    for f in range(n):
        e = np.asarray(exs[f], dtype=np.int64)
        if len(e) == 0:
            continue
        v = A[e, f]
        sd = v.std(ddof=1) if len(e) > 1 else np.nan
        stds[f] = sd if sd >= 1e-10 else 1.0
        Astd[e, f] = v - v.mean()
    Astd *= m * n / nobs
    U, S, Vt = np.linalg.svd(Astd)
    return np.sqrt(S[:k])[:, None] * U[:, :k].T, (np.sqrt(S[:k])[:, None] * Vt[:k]) * stds[None, :]


def svd_cell_models():
    """One model per sign assignment of the five singular pairs."""
    import itertools
    g0 = notebook_model()
    exs = [list(e) for e in g0.observed_examples]
    obs = [(i, j) for i, f in enumerate(g0.observed_features) for j in f]
    Xs, Ys = init_svd_numpy(g0.A, exs, 5, 100)
    for signs in itertools.product([1.0, -1.0], repeat=5):
        s = np.array(signs)[:, None]
        yield L.GLRM(g0.A, L.QuadLoss(), L.NonNegConstraint(), L.NonNegConstraint(), 5, observed_features=g0.observed_features,
                     observed_examples=g0.observed_examples, X=np.asfortranarray(s * Xs), Y=np.asfortranarray(s * Ys))


def check_svd_cell(fit, impute=None):
    matches = []
    for g in svd_cell_models():
        obj = np.array(fit(g))
        if abs(obj[1] - NOTEBOOK_SVD_HEAD[1]) <= 6e-6 * NOTEBOOK_SVD_HEAD[1]:
            matches.append((obj, None if impute is None else np.asarray(impute(g), dtype=float)))
        g.close()
    assert len(matches) == 1  # exactly one sign assignment takes the notebook's first step
    obj, Ahat = matches[0]
    if Ahat is not None:  # `impute(glrm)` of the next cell: X'Y for a real-valued QuadLoss model (src/impute_and_err.jl)
        assert Ahat.shape == (20, 10)
        np.testing.assert_allclose(Ahat[:, [0, 1, 2, 3, 7, 8, 9]], np.array(NOTEBOOK_IMPUTE), rtol=6e-6)
    assert len(obj) == 32 and obj[0] == np.inf
    for it, want in NOTEBOOK_SVD_OBJECTIVE.items():
        assert abs(obj[it] - want) <= 1e-9 * want, (it, obj[it], want)
    np.testing.assert_allclose(obj[1:10], NOTEBOOK_SVD_HEAD[1:], rtol=6e-6)
    np.testing.assert_allclose(obj[-10:], NOTEBOOK_SVD_TAIL, rtol=6e-6)


def test_oracle_reproduces_the_init_svd_cell():
    O.set_threads(1)
    check_svd_cell(lambda g: L.fit_b(g, L.ProxGradParams(), verbose=False, engine=O.oracle_api())[2].objective,
                   impute=lambda g: L.impute(g, engine=O.oracle_api()))


# ---------------------------------------------------------------------------------------------------------------------------
# sparse cell (:870-1030)
NOTEBOOK_SPARSE_ENTRIES = {  # (row, column) 1-based as printed -> value at six digits
    (1, 1): -0.39862, (2, 1): 0.865359, (3, 1): -1.52489, (4, 1): -0.0184348, (5, 1): -0.629805, (6, 1): 0.399439, (7, 1): -0.346628,
    (8, 1): 0.131743, (10, 1): 1.28734, (1, 2): 0.631291, (3, 2): -1.23373, (4, 2): -0.858585, (5, 9): 0.478738, (6, 9): 0.605134,
    (7, 9): 1.27697, (8, 9): -1.3962, (9, 9): 0.1171, (10, 9): 0.626685, (1, 10): 0.313382, (2, 10): 0.986252, (3, 10): 0.506609,
    (5, 10): 0.871816,
}
NOTEBOOK_SPARSE_OBJECTIVE = {10: 8.552659734332064, 20: 7.284079660640899, 30: 6.766039304968931}
# "Iteration 40" is the loop counter; one step between 30 and 40 was rejected ("obj went up ...") and adds no entry to ch.objective
NOTEBOOK_SPARSE_ITER40 = (39, 6.736723178681022)
NOTEBOOK_SPARSE_HEAD = [29.5239, 20.385, 16.1818, 13.0011, 11.2101, 10.1598, 9.56309, 9.19018, 8.9299, 8.72685]
NOTEBOOK_SPARSE_TAIL = [6.74587, 6.74447, 6.74427, 6.74359, 6.74218, 6.73993, 6.73672, 6.73367, 6.73297, 6.73297]


def sprandn(r, m, n, density):
    """SparseArrays.sprandn of Julia 1.1: sprand_IJ (a non-empty subsequence of rows per surviving column; randsubseq takes the plain
    `rand() <= p` loop for p > 0.15) and then one randn per stored entry, in column-major order.  Returns 0-based (I, J, V)."""
    import math
    Lg = math.log1p(-density)
    coldensity, colsparsity, iL = -math.expm1(m * Lg), math.exp(m * Lg), 1 / Lg
    assert density > 0.15 and coldensity > 0.15

    def randsubseq(count, p):
        return [i for i in range(1, count + 1) if r.rand() <= p]
    I, Jc = [], []
    for j in randsubseq(n, coldensity):
        kf = math.ceil(math.log(colsparsity + r.rand() * coldensity) * iL)
        ik = 1 if kf < 1 else m if kf > m else int(kf)
        rows = randsubseq(m - ik, density) + [m - ik + 1]
        I += rows
        Jc += [j] * len(rows)
    V = [r.randn() for _ in I]
    return np.array(I) - 1, np.array(Jc) - 1, np.array(V)


def sparse_cell_model(init="numpy", engine=None):
    import scipy.sparse as sp
    I, Jc, V = sprandn(J.MersenneTwister(1), 10, 10, 0.8)
    assert len(V) == 83  # "10x10 SparseMatrixCSC{Float64,Int64} with 83 stored entries"
    A = np.zeros((10, 10))
    A[I, Jc] = V
    for (i, j), v in NOTEBOOK_SPARSE_ENTRIES.items():
        assert abs(A[i - 1, j - 1] - v) <= 6e-6 * abs(v)
    g = L.GLRM(sp.csc_matrix(A), L.HuberLoss(), L.QuadReg(.1), L.QuadReg(.1), 5, rng=np.random.default_rng(0))
    if init == "numpy":
        X0, Y0 = init_svd_numpy(A, [list(e) for e in g.observed_examples], 5, 83)
        g.X[...], g.Y[...] = X0, Y0
    else:
        L.init_svd_(g, engine=engine)  # the package's init_svd! (subspace iteration on the resident lists)
    return g


def check_sparse_cell(obj, rtol):
    obj = np.array(obj)
    assert len(obj) == 43
    for it, want in list(NOTEBOOK_SPARSE_OBJECTIVE.items()) + [NOTEBOOK_SPARSE_ITER40]:
        assert abs(obj[it] - want) <= rtol * want, (it, obj[it], want)
    np.testing.assert_allclose(obj[:10], NOTEBOOK_SPARSE_HEAD, rtol=6e-6)
    np.testing.assert_allclose(obj[-10:], NOTEBOOK_SPARSE_TAIL, rtol=6e-6)


def test_oracle_reproduces_the_sparse_cell():
    O.set_threads(1)
    g = sparse_cell_model()
    X, Y, ch = L.fit_b(g, engine=O.oracle_api(), verbose=False)  # sparse A: SparseProxGradParams() by default (src/fit.jl)
    check_sparse_cell(ch.objective, rtol=1e-10)
    g.close()


def test_oracle_init_svd_reproduces_the_sparse_cell():
    O.set_threads(1)
    g = sparse_cell_model(init="engine", engine=O.oracle_api())
    X, Y, ch = L.fit_b(g, engine=O.oracle_api(), verbose=False)
    check_sparse_cell(ch.objective, rtol=1e-8)
    g.close()


@pytest.mark.gpu
def test_hip_engine_reproduces_the_init_svd_cell():
    check_svd_cell(lambda g: L.fit_b(g, L.HipProxGradParams(), verbose=False)[2].objective, impute=lambda g: L.impute(g))


@pytest.mark.gpu
def test_hip_engine_reproduces_the_sparse_cell():
    for init in ("numpy", "engine"):
        g = sparse_cell_model(init=init)
        X, Y, ch = L.fit_b(g, verbose=False)
        check_sparse_cell(ch.objective, rtol=1e-8)
        g.close()


@pytest.mark.gpu
def test_hip_engine_reproduces_the_notebook_trajectory():
    g = notebook_model()
    X, Y, ch = L.fit_b(g, L.HipProxGradParams(), verbose=False)
    check_against_the_notebook(X, Y, ch, rtol=1e-9)
    g.close()


@pytest.mark.gpu
def test_hip_engine_two_shards_reproduce_the_notebook_trajectory():
    g = notebook_model()
    X, Y, ch = L.fit_b(g, L.HipProxGradParams(ngpus=2, device_ids=[0, 0]), verbose=False)
    check_against_the_notebook(X, Y, ch, rtol=1e-9)
    g.close()
