"""GLRM(df, k, datatypes) (src/fit_dataframe.jl): level mapping, loss / regularizer choice, observations in column-major order,
and a fit of the resulting (categorical + ordinal + offset) model on the oracle engine."""
import numpy as np
import pandas as pd
import pytest

import extras as E
import lowrankmodels.jl_amd as L
import oracle as O


def table(rng, m=60):
    z = rng.standard_normal(m)
    df = pd.DataFrame({
        "height": 2 * z + 0.3 * rng.standard_normal(m),
        "smoker": np.where(z + 0.5 * rng.standard_normal(m) > 0, "yes", "no"),
        "grade": pd.Categorical(np.array(["low", "mid", "high", "top"])[np.clip(np.round(1.5 + z), 0, 3).astype(int)]).astype(object),
        "city": np.array(["ams", "ber", "cph"])[rng.integers(0, 3, m)],
    })
    df.loc[rng.random(m) < 0.2, "height"] = np.nan
    df.loc[rng.random(m) < 0.1, "city"] = None
    return df


def test_model_from_table():
    rng = np.random.default_rng(0)
    df = table(rng)
    g = E.glrm_from_dataframe(df, 3, ["real", "bool", "ord", "cat"], rng=rng)
    assert [type(l).__name__ for l in g.losses] == ["QuadLoss", "LogisticLoss", "MultinomialOrdinalLoss", "MultinomialLoss"]
    assert g.losses[2].max == 4 and g.losses[3].max == 3 and g.Y.shape == (3, 1 + 1 + 3 + 3)
    assert isinstance(g.ry[2], L.OrdinalReg) and isinstance(g.ry[0], L.lastentry_unpenalized) and isinstance(g.rx[0], L.lastentry1)
    # levels -> numbers: sorted levels; bools -> -1 / 1 (stored as false / true)
    grade_levels = sorted(set(df["grade"]))
    i = 7
    assert g.A[i, 2] == grade_levels.index(df["grade"][i]) + 1
    assert set(np.unique(g.A[:, 1])) == {-1.0, 1.0} and set(np.unique(g._colvals[g._colptr[1]:g._colptr[2]])) == {0.0, 1.0}
    # missing entries are not observed; lists are in column-major order (df_observations)
    nobs = int((~df.isna()).to_numpy().sum())
    assert int(g._rowptr[-1]) == nobs == int(g._colptr[-1])
    assert all(np.all(np.diff(g._colidx[g._rowptr[e]:g._rowptr[e + 1]]) > 0) for e in range(g.m))
    # prob_scale!: the QuadLoss column is scaled by 1 / (2 var)
    assert g.losses[0].scale == pytest.approx(1 / (2 * np.nanvar(g.A[:, 0], ddof=1)))
    X, Y, ch = L.fit_b(g, L.ProxGradParams(max_iter=150), verbose=False, engine=O.oracle_api())
    assert np.all(X[-1] == 1.0) and ch.objective[-1] < ch.objective[1]
    Ahat = L.impute(g, engine=O.oracle_api())
    assert set(np.unique(Ahat[:, 3])) <= {1.0, 2.0, 3.0} and set(np.unique(Ahat[:, 1])) <= {0.0, 1.0}
    assert set(np.unique(Ahat[:, 2])) <= {1.0, 2.0, 3.0, 4.0}


def test_argument_checks():
    df = table(np.random.default_rng(1), 20)
    with pytest.raises(ValueError):
        E.glrm_from_dataframe(df, 2, ["real", "bool", "ord"])
    with pytest.raises(ValueError):
        E.glrm_from_dataframe(df, 2, ["real", "bool", "ord", "text"])
    with pytest.raises(ValueError):
        E.glrm_from_dataframe(df, 2, ["real", "bool", "bool", "cat"])      # 'grade' has four levels
    with pytest.raises(ValueError):
        E.glrm_from_dataframe(df, 2, ["real", "real", "ord", "cat"])       # 'smoker' is not numeric
    g = E.glrm_from_dataframe(df, 2, ["real", "bool", "ord", "cat"], loss_map=E.robust_losses, offset=False, prob_scale=False)
    assert [type(l).__name__ for l in g.losses] == ["HuberLoss", "LogisticLoss", "BvSLoss", "OvALoss"] and isinstance(g.ry[2], L.OrdinalReg)
