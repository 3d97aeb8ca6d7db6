"""init_svd! at scale: glrm_hip_init_svd on the synthetic workload of BASELINE config 2 generated in HBM,
next to scipy's Arpack svds (what the reference calls, src/initialize.jl:121) on a host sample of the same generator.
    python tests/perf/bench_svd.py --m 1000000 --n 10000 --q 500 --k 32 [--cpu-rows 50000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lowrankmodels.jl_amd import _capi  # noqa: E402
from lowrankmodels.jl_amd.synth import DeviceWorkload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=1000000)
ap.add_argument("--n", type=int, default=10000)
ap.add_argument("--q", type=int, default=500)
ap.add_argument("--k", type=int, default=32)
ap.add_argument("--tol", type=float, default=1e-6)
ap.add_argument("--cpu-rows", type=int, default=50000)
a = ap.parse_args()
api = _capi.hip_api()
w = DeviceWorkload(a.m, a.n, a.k, a.q)
h = api.create(w.problem())
w.free_sources()
X, Y = np.zeros((a.k, a.m), order="F"), np.zeros((a.k, a.n), order="F")
api.init_svd(h, X, Y, max_iter=2, tol=a.tol)  # warm-up (allocations, first-touch)
t = time.time()
sv, iters = api.init_svd(h, X, Y, max_iter=100, tol=a.tol)
dt = time.time() - t
nnz = a.m * a.q
print(f"hip: m={a.m} n={a.n} k={a.k} nnz={nnz:.3g}: init_svd {dt:.2f} s, {iters} subspace iterations "
      f"({dt / iters * 1e3:.0f} ms each = 2 sparse products over {nnz:.3g} entries + 2 orthonormalizations); s1..s3 = {sv[:3]}", flush=True)
api.destroy(h)
if a.cpu_rows > 0:
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import oracle as O
    mr = min(a.cpu_rows, a.m)
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(mr, a.n, a.k, a.q)
    t = time.time()
    B = sp.csc_matrix((colvals, rowidx, colptr), shape=(mr, a.n))
    cnt = np.maximum(np.diff(colptr), 1)
    means = np.asarray(B.sum(axis=0)).ravel() / cnt
    B.data -= np.repeat(means, np.diff(colptr))
    B *= mr * a.n / len(colvals)
    U, S, Vt = spl.svds(B.tocsr(), k=a.k, tol=a.tol)
    dtc = time.time() - t
    print(f"cpu (scipy Arpack svds, {os.cpu_count()} hardware threads available): {mr} rows, nnz={len(colvals):.3g}: {dtc:.2f} s "
          f"-> {dtc / len(colvals) * 1e9:.1f} ns per observation vs {dt / nnz * 1e9:.3f} ns on the GPU")
