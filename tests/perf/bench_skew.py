"""Skewed row lengths (real ratings-style data): rows of similar length are handed to the lane groups of one wave (segperm), so a wave
does not wait for its one long row.   python tests/perf/bench_skew.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lowrankmodels.jl_amd as L
from lowrankmodels.jl_amd import _capi
from lowrankmodels.jl_amd.losses import pack_losses
from lowrankmodels.jl_amd.regularizers import pack_regs

rng = np.random.default_rng(0)
m, n, k = 300000, 4000, 32
p_row = np.clip(0.008 / rng.random(m) ** 0.7, 0.002, 0.5)          # heavy-tailed row densities
rows, cols = [], []
for b in range(0, m, 10000):
    mask = rng.random((min(10000, m - b), n)) < p_row[b:b + 10000, None]
    i, j = np.nonzero(mask)
    rows.append(i + b); cols.append(j)
I, J = np.concatenate(rows), np.concatenate(cols)
vals = rng.standard_normal(len(I))
rowptr = np.concatenate([[0], np.cumsum(np.bincount(I, minlength=m))]).astype(np.int64)
perm = np.argsort(J, kind="stable")
colptr = np.concatenate([[0], np.cumsum(np.bincount(J, minlength=n))]).astype(np.int64)
pa = _capi.ProblemArrays(m, n, k, rowptr, J.astype(np.int32), vals, colptr, I[perm].astype(np.int32), vals[perm],
                         pack_losses([L.QuadLoss()]), pack_regs([L.QuadReg(1.0)]), pack_regs([L.QuadReg(1.0)]))
lens = np.diff(rowptr)
print(f"nnz={len(I):.3g}, row lengths: mean {lens.mean():.0f}, median {np.median(lens):.0f}, max {lens.max()}", flush=True)
api = _capi.hip_api()
X0, Y0 = rng.standard_normal((k, m)) / 3, rng.standard_normal((k, n)) / 3
for label, env in (("segments by length", "1"), ("natural order", "0")):
    os.environ["GLRM_HIP_SEGPERM"] = env
    h = api.create(pa, profile=1, tiled=2)
    api.set_factors(h, np.asfortranarray(X0), np.asfortranarray(Y0)); api.reset_stepsizes(h, 1.0)
    for _ in range(2):
        api.step_x(h, 0.01); api.step_y(h, 0.01)
    api.synchronize(h); api.kernel_stats(h, reset=True)
    for _ in range(5):
        api.step_x(h, 0.01); api.step_y(h, 0.01)
    api.synchronize(h)
    st = api.kernel_stats(h)
    api.col_losses(h)
    print(f"{label:20s}: X half-step {st['ms_x'] / 5:.2f} ms, Y half-step {st['ms_y'] / 5:.2f} ms (families {st['tiled']})", flush=True)
    api.destroy(h)
