"""Driver-level fusion: time per fold of building the train / test models of a k-fold cross-validation
(a) from host arrays (upload 12 B per observation and view) vs (b) as glrm_hip_subset of the resident parent handle (1 tag byte
per observation and view), plus the fit time for scale.   python tests/perf/bench_cv.py --m 200000 --n 10000 --q 500 --k 32"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lowrankmodels.jl_amd as L  # noqa: E402
from lowrankmodels.jl_amd import _capi  # noqa: E402
import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=200000)
ap.add_argument("--n", type=int, default=10000)
ap.add_argument("--q", type=int, default=500)
ap.add_argument("--k", type=int, default=32)
ap.add_argument("--folds", type=int, default=5)
a = ap.parse_args()
rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(a.m, a.n, a.k, a.q)
from lowrankmodels.jl_amd.losses import pack_losses  # noqa: E402
from lowrankmodels.jl_amd.regularizers import pack_regs  # noqa: E402
pa = _capi.ProblemArrays(a.m, a.n, a.k, rowptr, colidx, rowvals, colptr, rowidx, colvals, pack_losses([L.QuadLoss()]),
                         pack_regs([L.QuadReg(1.0)]), pack_regs([L.QuadReg(1.0)]))
nnz = int(pa.rowptr[-1])
api = _capi.hip_api()
t = time.time(); h = api.create(pa); api.synchronize(h); t_create = time.time() - t
rng = np.random.default_rng(0)
tags = rng.integers(0, a.folds, nnz).astype(np.uint8)
I = np.repeat(np.arange(a.m), np.diff(pa.rowptr)); J = pa.colidx.astype(np.int64)
from lowrankmodels.jl_amd.crossval import _stable_order_by_column  # noqa: E402
t = time.time(); perm = _stable_order_by_column(pa.rowptr, pa.colidx, a.n); ctags = tags[perm]; t_perm = time.time() - t
p = L.ProxGradParams(max_iter=10, abs_tol=0, rel_tol=0)
t_sub = t_host = t_fit = 0.0
for f in range(a.folds):
    t = time.time(); htr = api.subset(h, tags, ctags, f, True); hte = api.subset(h, tags, ctags, f, False); api.synchronize(hte); t_sub += time.time() - t
    X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
    t = time.time(); obj, _ = api.fit(htr, p, X, Y); t_fit += time.time() - t
    te = api.objective(hte, X, Y, False) / api.kernel_stats(hte)["nnz_rows"]
    api.destroy(htr); api.destroy(hte)
    if f == 0:  # the unfused alternative: compact on the host, upload both children
        t = time.time()
        keep = tags != f
        def child(kr):
            kc = kr[perm]
            rp = np.concatenate([[0], np.cumsum(np.bincount(I[kr], minlength=a.m))]); cp = np.concatenate([[0], np.cumsum(np.bincount(J[perm][kc], minlength=a.n))])
            return _capi.ProblemArrays(a.m, a.n, a.k, rp.astype(np.int64), pa.colidx[kr], pa.rowvals[kr], cp.astype(np.int64), pa.rowidx[kc], pa.colvals[kc], pa.losses, pa.rx, pa.ry)
        h1 = api.create(child(keep)); h2 = api.create(child(~keep)); api.synchronize(h2)
        t_host = time.time() - t
        api.destroy(h1); api.destroy(h2)
print(f"nnz={nnz:.3g}: parent create (host arrays -> device) {t_create:.2f} s; fold split on the device {t_sub / a.folds:.3f} s per fold "
      f"(+ {t_perm:.2f} s once for the column-order tags); fold split on the host + upload {t_host:.2f} s per fold; "
      f"10-iteration fit {t_fit / a.folds:.2f} s per fold; last test error {te:.4g}")
