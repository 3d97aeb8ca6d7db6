"""Where does a fuzz model leave the oracle?  Steps the HIP engine and the oracle side by side (step-level API, same start) and prints,
after every half-step, the largest deviations per row of X / per column block of Y with the loss kinds involved.
python tests/perf/dbg_fuzz_steps.py 12 [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, oracle as O
import importlib.util
from lowrankmodels.jl_amd import _capi
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
seed = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g, p = fz.random_model(seed)
pa = g.problem_arrays()
X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
ys = pa.ystart
apis = (O.oracle_api(), _capi.hip_api())
hs = [a.create(pa) for a in apis]
for a, h in zip(apis, hs):
    a.set_factors(h, X0, Y0); a.reset_stepsizes(h, p.stepsize)
def get(a, h):
    X, Y = np.zeros_like(X0), np.zeros_like(Y0); a.get_factors(h, X, Y); return X, Y
names = [type(l).__name__ for l in g.losses]
for it in range(iters):
    for which in ("x", "y"):
        for a, h in zip(apis, hs):
            (a.step_x if which == "x" else a.step_y)(h, p.min_stepsize)
        (Xc, Yc), (Xg, Yg) = get(apis[0], hs[0]), get(apis[1], hs[1])
        with np.errstate(all="ignore"):
            ex = np.max(np.abs(Xg - Xc) / (np.abs(Xc) + 1e-300), axis=0)
            ey = np.max(np.abs(Yg - Yc) / (np.abs(Yc) + 1e-300), axis=0)
        print(f"iter {it} after step_{which}: X fro {cases.fro_err(Xg, Xc):.2e} Y fro {cases.fro_err(Yg, Yc):.2e} |X|max {np.abs(Xc).max():.3g} |Y|max {np.abs(Yc).max():.3g}")
        for e in np.argsort(-ex)[:3]:
            if ex[e] > 1e-9:
                cols = pa.colidx[pa.rowptr[e]:pa.rowptr[e + 1]]
                print(f"     row {e}: rel {ex[e]:.2e} |x| {np.abs(Xc[:, e]).max():.3g} vs {np.abs(Xg[:, e]).max():.3g}; kinds {sorted(set(names[c] for c in cols))}")
        for v in np.argsort(-ey)[:3]:
            if ey[v] > 1e-9:
                f = int(np.searchsorted(ys, v, side='right') - 1)
                print(f"     Y vector {v} (column {f}, {names[f]}): rel {ey[v]:.2e} |y| {np.abs(Yc[:, v]).max():.3g} vs {np.abs(Yg[:, v]).max():.3g}")
for a, h in zip(apis, hs):
    a.destroy(h)
