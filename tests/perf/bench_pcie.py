"""PCIe-inclusive rate of the boundary when the caller hands over HOST arrays (what a Julia `fit!` does): seconds of glrm_hip_create
(copies both Omega views + values once per handle) and of a 50-iteration glrm_hip_fit (after 5 iterations that leave the random start behind) (moves X and Y in and out), against the resident
rate bench.py reports.  C2 recipe at 1/5 of the rows: 200 000 x 10 000, rank 32, 1e8 observations (2.4 GB of lists)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as O
from lowrankmodels.jl_amd import _capi, synth
import lowrankmodels.jl_amd as L

m, n, k, q = 200_000, 10_000, 32, 500
t0 = time.time()
rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, seed=1, value_model=0, loss_mix=0)
reg = np.array([(1, 0, 1.0)], dtype=_capi.REG_DTYPE)  # QuadReg(1.0): (kind, wrap, scale)
pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, synth.loss_table(n, 0), reg, reg)
gen = time.time() - t0
api = _capi.hip_api()
nbytes = sum(a.nbytes for a in (rowptr, colidx, rowvals, colptr, rowidx, colvals))
t0 = time.time(); h = api.create(pa, profile=1); api.synchronize(h); t_create = time.time() - t0
p = L.ProxGradParams(max_iter=50, abs_tol=-1e300, rel_tol=-1e300)
X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
api.fit(h, L.ProxGradParams(max_iter=5, abs_tol=-1e300, rel_tol=-1e300), X, Y)  # past the long line searches of a random start
t0 = time.time(); obj, sec = api.fit(h, p, X, Y); t_fit = time.time() - t0
iters = len(obj) - 1
upd = 2.0 * len(colidx) * iters
print(json.dumps({"observations": int(len(colidx)), "list_bytes": int(nbytes), "host_generate_s": round(gen, 2), "create_s": round(t_create, 3),
                  "create_GBps": round(nbytes / t_create / 1e9, 1), "fit_iterations": iters, "fit_s": round(t_fit, 3),
                  "resident_updates_per_s": upd / sec[-1] if sec[-1] > 0 else None, "fit_call_updates_per_s": upd / t_fit,
                  "create_plus_fit_updates_per_s": upd / (t_create + t_fit)}))
print(api.kernel_stats(h))
api.destroy(h)
