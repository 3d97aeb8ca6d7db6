"""BASELINE.md tier T2: the reference's own cost model (dense XY = X'Y, full-row / full-column objectives, src/algorithms/proxgrad.jl
:65-66,135-143; src/evaluate_fit.jl:24-55) timed beside the observed-only oracle and the HIP engine at C1 and at 100k x 1k, k=32, 5 %
observed.  All three produce the same numbers (tests/test_golden.py::test_dense_faithful_mode_is_the_same_arithmetic); only the cost
differs.   python tests/perf/bench_faithful.py [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lowrankmodels.jl_amd as L
from lowrankmodels.jl_amd import _capi
import oracle as O

threads = int(sys.argv[1]) if len(sys.argv) > 1 else O.usable_cores()
O.set_threads(threads)
try:
    hip = _capi.hip_api()
    import torch
    hip = hip if torch.cuda.is_available() else None
except Exception:
    hip = None
cpu, lib = O.oracle_api(), O.oracle_lib()
reg = np.array([(1, 0, 0.1)], dtype=_capi.REG_DTYPE)
one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
for (m, n, k, q, iters) in [(100, 100, 5, 100, 50), (100_000, 1000, 32, 50, 5)]:
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    p = L.ProxGradParams(max_iter=iters, abs_tol=0.0, rel_tol=-1.0)
    res = {}
    for name in ("dense-faithful oracle", "observed-only oracle", "hip engine"):
        eng = hip if name.startswith("hip") else cpu
        if eng is None:
            continue
        h = eng.create(pa)
        if name.startswith("dense"):
            assert lib.glrm_cpu_set_dense_faithful(h, 1) == 0
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        eng.fit(h, L.ProxGradParams(max_iter=1), X.copy(order="F"), Y.copy(order="F"))
        t = time.time(); obj, _ = eng.fit(h, p, X, Y); dt = time.time() - t
        eng.destroy(h)
        res[name] = obj
        ups = (len(obj) - 1) * 2 * int(rowptr[-1]) / dt
        print(f"{m}x{n} k={k} nnz={int(rowptr[-1])} threads={threads}: {name:22s} {1e3 * dt / (len(obj) - 1):10.3f} ms per outer iteration, {ups:.3g} updates/s, final objective {obj[-1]:.6e}")
    a = res["dense-faithful oracle"]
    for name, o in res.items():
        print(f"   max relative objective difference {name} vs dense-faithful: {np.max(np.abs(o - a) / np.abs(a)):.2e}")
