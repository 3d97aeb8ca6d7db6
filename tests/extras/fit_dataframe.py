"""TEST-SIDE helper (not part of the product: SURVEY.md section 2 marks the DataFrame constructor out of scope; it lives here only
because the replayed reference scripts -- test/hello_world.jl -- build their models with it).

GLRM(df, k, datatypes) (reference: src/fit_dataframe.jl:12-79,83-205): a model for a table with :real / :bool / :ord / :cat
columns -- levels mapped to numbers, one loss per column from a loss map, OrdinalReg on multi-dimensional ordinal columns, an
unpenalized offset, prob_scale!.  The table is a pandas DataFrame (or anything pandas.DataFrame accepts)."""
from __future__ import annotations

import copy as _copy

import numpy as np

from lowrankmodels.jl_amd.glrm import GLRM
from lowrankmodels.jl_amd.losses import BvSLoss, HuberLoss, LogisticLoss, MultinomialLoss, MultinomialOrdinalLoss, OrdisticLoss, OvALoss, QuadLoss
from lowrankmodels.jl_amd.regularizers import OrdinalReg, QuadReg
from .prob_scale import prob_scale_

probabilistic_losses = {"real": QuadLoss, "bool": LogisticLoss, "ord": MultinomialOrdinalLoss, "cat": MultinomialLoss}
robust_losses = {"real": HuberLoss, "bool": LogisticLoss, "ord": BvSLoss, "cat": OvALoss}


def map_to_numbers(col, datatype):
    """map_to_numbers!(df, j, datatype) (:83-120): :real stays, :bool levels -> -1 / 1, :cat / :ord levels -> 1..nlevels (sorted)."""
    import pandas as pd
    miss = pd.isna(col).to_numpy()
    out = np.full(len(col), np.nan)
    if datatype == "real":
        try:
            out[~miss] = np.asarray(col[~miss], dtype=float)
        except (TypeError, ValueError):
            raise ValueError("column contains non-numerical values")
        return out
    levels = sorted(set(col[~miss].tolist()))
    if datatype == "bool":
        if len(levels) > 2:
            raise ValueError(f"Boolean variable should have at most two levels; instead, got:\n{levels}")
        colmap = dict(zip(levels, [-1, 1][:len(levels)]))
    elif datatype in ("cat", "ord"):
        colmap = dict(zip(levels, range(1, len(levels) + 1)))
    else:
        raise ValueError(f"datatype {datatype} not recognized")
    out[~miss] = [colmap[v] for v in col[~miss].tolist()]
    return out


def pick_loss(losstype, col):
    """pick_loss (:178-205).  Losses that take the number of levels get max(col); the reference only defines that for
    MultinomialLoss / MultinomialOrdinalLoss (its robust map's BvSLoss() / OvALoss() do not construct) -- here they get it too."""
    obs = col[~np.isnan(col)]
    if losstype is LogisticLoss:
        if not np.all(np.isin(obs, (-1, 1))):
            raise ValueError("LogisticLoss can only be used on data taking values in {-1, 1}")
        return LogisticLoss()
    if losstype in (MultinomialLoss, MultinomialOrdinalLoss, BvSLoss, OvALoss, OrdisticLoss):
        if not (np.all(obs >= 1) and np.all(obs == np.floor(obs))):
            raise ValueError(f"{losstype.__name__} can only be used on data taking positive integer values")
        return losstype(int(obs.max()) if len(obs) else 2)
    return losstype()


def observations(A):
    """df_observations (:208-219): (i, j) of the non-missing entries in column-major order."""
    J, I = np.nonzero(~np.isnan(A).T)
    return I, J


def glrm_from_dataframe(df, k, datatypes, *, loss_map=None, rx=None, ry=None, offset=True, scale=False, prob_scale=True,
                        transform_data_to_numbers=True, **kwargs):
    """GLRM(df, k, datatypes; loss_map, rx, ry, offset, scale, prob_scale, transform_data_to_numbers)."""
    import pandas as pd
    df = pd.DataFrame(df)
    loss_map = probabilistic_losses if loss_map is None else loss_map
    rx = QuadReg(.01) if rx is None else rx
    ry = QuadReg(.01) if ry is None else ry
    datatypes = [str(d).lstrip(":") for d in datatypes]
    if df.shape[1] != len(datatypes):
        raise ValueError("third argument (datatypes) must have one entry for each column of data frame.")
    for dt in datatypes:
        if dt not in loss_map:
            raise ValueError(f"data types must be either :real, :bool, :ord, or :cat, not {dt}")
    m, n = df.shape
    A = np.full((m, n), np.nan)
    losses = []
    for j, dt in enumerate(datatypes):
        col = df.iloc[:, j]
        A[:, j] = map_to_numbers(col, dt) if transform_data_to_numbers else np.asarray(col, dtype=float)
        losses.append(pick_loss(loss_map[dt], A[:, j]))
    obs = observations(A)
    rys = [OrdinalReg(_copy.copy(ry)) if isinstance(l, (MultinomialOrdinalLoss, BvSLoss, OrdisticLoss)) else _copy.copy(ry) for l in losses]
    glrm = GLRM(A, losses, rx, rys, k, obs=obs, offset=offset, scale=scale, **kwargs)
    if prob_scale:
        prob_scale_(glrm)
    return glrm
