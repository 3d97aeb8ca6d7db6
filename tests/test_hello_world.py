"""The reference's smoke test test/hello_world.jl, line for line where the accelerated path covers it: a mixed model (real, bool,
ordinal and categorical losses with random scales, per-row regularizers), fully observed and with entries observed with
replacement, the table constructor GLRM(DataFrame(A), 3, data_types), impute.  (KSparseConstraint is outside the descriptor
table and left out; sample / sample_missing are post-fit sampling, out of scope.)  Runs on the oracle here and on the HIP engine
under -m gpu."""
import numpy as np
import pandas as pd
import pytest

import extras as E
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi


def hello_world(api, rng):
    ncat, nord = 4, 5
    real_losses = [L.QuadLoss(), L.HuberLoss()]
    bool_losses = [L.HingeLoss()]
    ordinal_losses = [L.OrdinalHingeLoss(1, int(rng.integers(3, nord + 1))), L.BvSLoss(int(rng.integers(3, nord + 1)))]
    categorical_losses = [L.MultinomialLoss(int(rng.integers(3, ncat + 1))), L.OvALoss(int(rng.integers(3, ncat + 1)))]
    losses = real_losses + bool_losses + ordinal_losses + categorical_losses
    data_types = ["real"] * 2 + ["bool"] + ["ord"] * 2 + ["cat"] * 2
    for lo in losses:
        lo.mul_(float(rng.random()))
    regularizers = [L.QuadReg(), L.OneReg(5), L.NonNegConstraint()] + [L.QuadReg() for _ in range(10)]
    m, n = len(regularizers), len(losses)
    A = np.column_stack([rng.random((m, 2)), rng.random((m, 1)) < 0.5, rng.integers(1, 4, (m, 2)), rng.integers(1, 4, (m, 2))]).astype(float)
    p = L.HipProxGradParams() if api is None else L.ProxGradParams()
    out = []
    g = L.GLRM(A, losses, regularizers, L.QuadReg(), 2, rng=rng)
    out.append(L.fit_b(g, p, verbose=False, engine=api)[2])                                    # "successfully fit matrix"
    omega = (rng.integers(0, m, 5 * max(m, n)), rng.integers(0, n, 5 * max(m, n)))               # with replacement
    g = L.GLRM(A, losses, regularizers, L.QuadReg(), 2, obs=omega, rng=rng)
    out.append(L.fit_b(g, p, verbose=False, engine=api)[2])                                    # "... with some entries unobserved"
    S = rng.standard_normal((10, 10))
    S[rng.random((10, 10)) < 0.5] = np.nan
    I, J = np.nonzero(~np.isnan(S))
    g = L.GLRM(S, L.QuadLoss(), L.QuadReg(), L.QuadReg(), 2, obs=(I, J), rng=rng)                # explicitly encoded missing entries
    out.append(L.fit_b(g, p, verbose=False, engine=api)[2])
    g = E.glrm_from_dataframe(pd.DataFrame(A), 3, data_types, rng=rng)                            # "successfully fit dataframe"
    out.append(L.fit_b(g, p, verbose=False, engine=api)[2])
    Ahat = L.impute(g, engine=api)                                                                # "successfully imputed entries"
    assert Ahat.shape == A.shape and np.all(np.isfinite(Ahat))
    return out


def test_hello_world_on_the_oracle():
    chs = hello_world(O.oracle_api(), np.random.default_rng(0))
    assert all(np.isfinite(ch.objective[-1]) and ch.objective[-1] <= ch.objective[1] for ch in chs)


@pytest.mark.gpu
def test_hello_world_on_the_gpu_matches_the_oracle():
    a = hello_world(None, np.random.default_rng(0))
    b = hello_world(O.oracle_api(), np.random.default_rng(0))
    for cg, cc in zip(a, b):
        n = min(len(cg.objective), len(cc.objective), 12)       # the first iterations, before rounding can flip a decision
        ref = np.array(cc.objective[:n]); got = np.array(cg.objective[:n])
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(got), fin) and np.allclose(got[fin], ref[fin], rtol=1e-5)
