"""Recorded objectives of a fuzz seed on both engines (whole-fit entry point).  python tests/perf/dbg_obj.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, oracle as O
import importlib.util
import lowrankmodels.jl_amd as L
from lowrankmodels.jl_amd import _capi
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
seed = int(sys.argv[1])
g, p = fz.random_model(seed)
pa = g.problem_arrays()
X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
T = fz.well_conditioned_prefix(pa, X0, Y0, p, seed)
q = L.ProxGradParams(p.stepsize, max_iter=min(p.max_iter, T), inner_iter=p.inner_iter_X, abs_tol=0.0, rel_tol=-1.0)
o_c, X_c, Y_c, _ = cases.run_engine(O.oracle_api(), pa, X0, Y0, q)
o_g, X_g, Y_g, _ = cases.run_engine(_capi.hip_api(), pa, X0, Y0, q)
np.set_printoptions(precision=17, linewidth=200)
print("stable prefix", T); print("cpu", o_c); print("hip", o_g)
print("X fro", cases.fro_err(X_g, X_c), "Y fro", cases.fro_err(Y_g, Y_c))
for api, name in ((O.oracle_api(), "cpu"), (_capi.hip_api(), "hip")):
    h = api.create(pa)
    try:
        print(name, "objective(X_c, Y_c): with reg", api.objective(h, X_c, Y_c, True), "without", api.objective(h, X_c, Y_c, False),
              "| at start: with", api.objective(h, X0, Y0, True), "without", api.objective(h, X0, Y0, False))
    finally:
        api.destroy(h)
