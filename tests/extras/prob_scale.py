"""TEST-SIDE helper (out of scope for the product): prob_scale!(glrm), used by the DataFrame constructor (src/fit_dataframe.jl:66-67)."""
import numpy as np

import lowrankmodels.jl_amd.losses as _l
from .scaling import _observed_values, avgerror


def prob_scale_(glrm, columns_to_scale=None, TOL=1e-12):
    """prob_scale!(glrm): -log-likelihood scaling of Quad / Huber columns (src/modify_glrm.jl:60-82; TOL = 1e-12 is the module constant of
    src/regularizers.jl:25; the 1e-3 values there are keyword defaults of the MNL ordinal rules, not this constant)."""
    cols = range(glrm.n) if columns_to_scale is None else columns_to_scale
    A = np.asarray(glrm.A, dtype=float)
    for i in cols:
        l = glrm.losses[i]
        nomissing = _observed_values(glrm, i)
        if type(l) is _l.QuadLoss and len(nomissing) > 0:
            col = A[:, i][~np.isnan(A[:, i])]                 # var(skipmissing(glrm.A[:,i]))
            v = float(np.var(col, ddof=1)) if len(col) > 1 else 0.0
            if v > TOL:
                l.mul_(1 / (2 * v))
        elif type(l) is _l.HuberLoss and len(nomissing) > 0:
            v = avgerror(l, A[:, i][~np.isnan(A[:, i])])      # avgerror collects skipmissing(a)
            if v > TOL:
                l.mul_(1 / (2 * v))
        else:
            l.mul_(1)
    glrm.close()
    return glrm
