"""Latency of small fits (BASELINE config 1 shape and friends): wall time per outer iteration of glrm_hip_fit, where launches and the
per-iteration objective read-back dominate.   python tests/perf/bench_small.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lowrankmodels.jl_amd as L
from lowrankmodels.jl_amd import _capi
import oracle as O

api = _capi.hip_api()
for (m, n, k, dens) in [(100, 100, 5, 1.0), (1000, 500, 8, 0.2), (20000, 2000, 16, 0.05), (100000, 5000, 32, 0.02), (300000, 3000, 32, 0.05), (1000000, 2000, 32, 0.05)]:
    rng = np.random.default_rng(0)
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n))
    obs = None if dens >= 1 else np.nonzero(rng.random((m, n)) < dens)
    g = L.GLRM(A, L.QuadLoss(), L.QuadReg(0.1), L.QuadReg(0.1), k, obs=obs, rng=rng)
    pa = g.problem_arrays(dense=False)
    p = L.ProxGradParams(max_iter=60, abs_tol=0.0, rel_tol=-1.0)
    for name, eng in (("hip auto", api), ("hip gather", api), ("hip tiled", api), ("hip graph on", api), ("hip graph off", api)):
        if name.startswith("cpu"):
            O.set_threads(min(16, os.cpu_count()))
        if name.startswith("hip graph"):  # no per-launch events: the hipGraph of one outer iteration is eligible (gather sweeps)
            os.environ["GLRM_HIP_GRAPH"] = "1" if name.endswith("on") else "0"
            h = eng.create(pa, tiled=1)
        else:
            h = eng.create(pa, profile=1, tiled={"hip auto": 0, "hip gather": 1, "hip tiled": 2}[name]) if name.startswith("hip") else eng.create(pa)
        X, Y = np.array(g.X, order="F"), np.array(g.Y, order="F")
        eng.fit(h, L.ProxGradParams(max_iter=3), X.copy(order="F"), Y.copy(order="F"))  # warm-up
        if name.startswith("hip") and not name.startswith("hip graph"):
            eng.kernel_stats(h, reset=True)
        t = time.time(); obj, sec = eng.fit(h, p, X, Y); dt = time.time() - t
        extra = ""
        if name.startswith("hip") and not name.startswith("hip graph"):
            st = eng.kernel_stats(h)
            extra = f"  [X {1e3 * st['ms_x'] / (len(obj) - 1):.0f} us, Y {1e3 * st['ms_y'] / (len(obj) - 1):.0f} us per iteration, families {st['tiled']}]"
        eng.destroy(h)
        print(f"{m}x{n} k={k} nnz={int(pa.rowptr[-1])}: {name:10s} {1e6 * dt / (len(obj) - 1):8.1f} us per outer iteration ({len(obj) - 1} iterations, {dt * 1e3:.1f} ms){extra}")
