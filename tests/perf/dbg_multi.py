"""Debug helper: one X half-step and one Y half-step on the oracle and on the HIP engine from the same state.
    python tests/perf/dbg_multi.py <case name | custom:LOSS,LOSS,...>   e.g. custom:ova3,bvs5,quad"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, oracle as O
import lowrankmodels.jl_amd as L
from lowrankmodels.jl_amd import _capi
name = sys.argv[1] if len(sys.argv) > 1 else "categorical_mix"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if name.startswith("custom:"):
    mk = {"ova": lambda d: L.OvALoss(d), "ovah": lambda d: L.OvALoss(d, bin_loss=L.HingeLoss()), "bvs": lambda d: L.BvSLoss(d),
          "bvsh": lambda d: L.BvSLoss(d, bin_loss=L.HingeLoss()), "mnl": lambda d: L.MultinomialLoss(d), "ord": lambda d: L.OrdisticLoss(d),
          "mno": lambda d: L.MultinomialOrdinalLoss(d), "quad": lambda d: L.QuadLoss(), "log": lambda d: L.LogisticLoss()}
    losses = []
    for tok in name[7:].split(","):
        nm = tok.rstrip("0123456789"); d = int(tok[len(nm):] or 0)
        losses.append(mk[nm](d))
    kwargs, p = cases._multidim_data(np.random.default_rng(5), 28, k, losses, L.QuadReg(0.1), L.QuadReg(0.2), L.ProxGradParams(max_iter=12))
else:
    kwargs, p = cases.build_multidim_case(name)
g = L.GLRM(**kwargs)
pa = g.problem_arrays()
ys = pa.ystart
res = {}
for label, api in (("cpu", O.oracle_api()), ("hip", _capi.hip_api())):
    h = api.create(pa)
    X, Y = np.array(kwargs["X"], order="F"), np.array(kwargs["Y"], order="F")
    api.set_factors(h, X, Y); api.reset_stepsizes(h, 1.0)
    api.step_x(h, 0.01); api.get_factors(h, X, Y); X1 = X.copy()
    api.step_y(h, 0.01); api.get_factors(h, X, Y); Y1 = Y.copy()
    res[label] = (X1, Y1)
    api.destroy(h)
dx = np.abs(res["cpu"][0] - res["hip"][0]).max(axis=0)
dy = np.abs(res["cpu"][1] - res["hip"][1]).max(axis=0)
print(name, "k", k, "X max diff %.3g" % dx.max(), "Y max diff %.3g" % dy.max())
if len(sys.argv) > 3:
    for e in range(g.m):
        cols = list(g._colidx[g._rowptr[e]:g._rowptr[e + 1]])
        vals = list(g._rowvals[g._rowptr[e]:g._rowptr[e + 1]])
        print(e, "obs cols", cols, "vals", vals, "dx %.3g" % dx[e])
