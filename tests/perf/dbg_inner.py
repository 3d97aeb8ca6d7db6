"""Seed with inner iterations: whole-fit entry point against the step-level API on both engines.  python tests/perf/dbg_inner.py SEED [ITERS]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, oracle as O
import importlib.util
import lowrankmodels.jl_amd as L
from lowrankmodels.jl_amd import _capi
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
seed = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g, p = fz.random_model(seed)
pa = g.problem_arrays()
X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
inner = p.inner_iter_X
q = L.ProxGradParams(p.stepsize, max_iter=iters, inner_iter=inner, abs_tol=0.0, rel_tol=-1.0)
names = [type(l).__name__ for l in g.losses]
ys = pa.ystart
def fit(api):
    return cases.run_engine(api, pa, X0, Y0, q)
def steps(api):
    h = api.create(pa)
    try:
        api.set_factors(h, X0, Y0); api.reset_stepsizes(h, q.stepsize)
        for _ in range(iters):
            for _ in range(inner): api.step_x(h, q.min_stepsize)
            for _ in range(inner): api.step_y(h, q.min_stepsize)
        X, Y = np.zeros_like(X0), np.zeros_like(Y0); api.get_factors(h, X, Y)
        st = api.kernel_stats(h)
    finally:
        api.destroy(h)
    return None, X, Y, st
res = {"cpu fit": fit(O.oracle_api()), "hip fit": fit(_capi.hip_api()), "cpu steps": steps(O.oracle_api()), "hip steps": steps(_capi.hip_api())}
def cmp(a, b):
    (_, Xa, Ya, sa), (_, Xb, Yb, sb) = res[a], res[b]
    with np.errstate(all="ignore"):
        ey = np.max(np.abs(Ya - Yb) / (np.abs(Yb) + 1e-300), axis=0); ex = np.max(np.abs(Xa - Xb) / (np.abs(Xb) + 1e-300), axis=0)
    print(f"{a} vs {b}: X fro {cases.fro_err(Xa, Xb):.2e} Y fro {cases.fro_err(Ya, Yb):.2e}; trials x/y {sa['trials_x']}/{sa['trials_y']} vs {sb['trials_x']}/{sb['trials_y']}; accepts {sa['accepts_x']}/{sa['accepts_y']} vs {sb['accepts_x']}/{sb['accepts_y']}")
    for v in np.argsort(-ey)[:4]:
        if ey[v] > 1e-9:
            f = int(np.searchsorted(ys, v, side='right') - 1)
            print(f"     Y vector {v} (column {f}, {names[f]}, ry {type(g.ry[f]).__name__}/{type(getattr(g.ry[f], 'r', None)).__name__}): rel {ey[v]:.2e}  a {Ya[:, v][:4]} b {Yb[:, v][:4]}")
    for e in np.argsort(-ex)[:3]:
        if ex[e] > 1e-9: print(f"     row {e}: rel {ex[e]:.2e}")
print("objective cpu", res["cpu fit"][0]); print("objective hip", res["hip fit"][0])
cmp("hip fit", "cpu fit"); cmp("hip steps", "cpu steps"); cmp("cpu steps", "cpu fit"); cmp("hip steps", "hip fit")
