"""Mid-size problem (gather sweeps) with Zipf column popularity: how the waves-per-column choice copes with a few very long
columns.   python tests/perf/bench_skew_cols.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lowrankmodels.jl_amd as L
from lowrankmodels.jl_amd import _capi
from lowrankmodels.jl_amd.losses import pack_losses
from lowrankmodels.jl_amd.regularizers import pack_regs

rng = np.random.default_rng(0)
m, n, k = 100000, 5000, 32
p_col = 1.0 / np.arange(1, n + 1) ** 0.9
p_col = np.minimum(p_col / p_col.mean() * 0.02, 0.9)
rows, cols = [], []
for b in range(0, m, 10000):
    i, j = np.nonzero(rng.random((10000, n)) < p_col[None, :])
    rows.append(i + b); cols.append(j)
I, J = np.concatenate(rows), np.concatenate(cols)
vals = rng.standard_normal(len(I))
rowptr = np.concatenate([[0], np.cumsum(np.bincount(I, minlength=m))]).astype(np.int64)
perm = np.argsort(J, kind="stable")
colptr = np.concatenate([[0], np.cumsum(np.bincount(J, minlength=n))]).astype(np.int64)
pa = _capi.ProblemArrays(m, n, k, rowptr, J.astype(np.int32), vals, colptr, I[perm].astype(np.int32), vals[perm],
                         pack_losses([L.QuadLoss()]), pack_regs([L.QuadReg(1.0)]), pack_regs([L.QuadReg(1.0)]))
cl = np.diff(colptr)
print(f"nnz={len(I):.3g}, column lengths: mean {cl.mean():.0f}, median {np.median(cl):.0f}, max {cl.max()}", flush=True)
api = _capi.hip_api()
X0, Y0 = rng.standard_normal((k, m)) / 3, rng.standard_normal((k, n)) / 3
for label, kw in (("auto", {}), ("gather, auto waves", dict(tiled=1)), ("gather, 8 waves per column", dict(tiled=1, waves_col=8)), ("gather, 1 wave per column", dict(tiled=1, waves_col=1)),
                  ("tiled", dict(tiled=2))):
    h = api.create(pa, profile=1, **kw)
    api.set_factors(h, np.asfortranarray(X0), np.asfortranarray(Y0)); api.reset_stepsizes(h, 1.0)
    for _ in range(2):
        api.step_x(h, 0.01); api.step_y(h, 0.01)
    api.synchronize(h); api.kernel_stats(h, reset=True)
    for _ in range(5):
        api.step_x(h, 0.01); api.step_y(h, 0.01)
    api.synchronize(h)
    st = api.kernel_stats(h)
    print(f"{label:28s}: X half-step {st['ms_x'] / 5:.2f} ms, Y half-step {st['ms_y'] / 5:.2f} ms (families {st['tiled']}, waves_col {st['waves_col']})", flush=True)
    api.destroy(h)
