"""Three ragged shards on one device against the single handle for a fuzz seed.  python tests/perf/dbg_shards.py SEED"""
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
np.set_printoptions(precision=17, linewidth=220)
o_g, X_g, Y_g, _ = cases.run_engine(_capi.hip_api(), pa, X0, Y0, q)
rb, cb = [0, g.m // 5, g.m // 2 + 1, g.m], [0, g.n // 4 + 1, g.n // 2, g.n]
print("row bounds", rb, "column bounds", cb, "losses", [type(l).__name__ for l in g.losses], "ry", [type(r).__name__ for r in g.ry])
o_s, X_s, Y_s, _ = cases.run_shards_on_one_device(_capi.hip_api(), pa, X0, Y0, q, rb, cb)
print("single", o_g[1:]); print("shards", o_s)
with np.errstate(all="ignore"):
    print("X equal", np.array_equal(X_s, X_g), "Y equal", np.array_equal(Y_s, Y_g), "max |dX|", np.nanmax(np.abs(X_s - X_g)), "max |dY|", np.nanmax(np.abs(Y_s - Y_g)))
    bad = np.argwhere(~((Y_s == Y_g) | (np.isnan(Y_s) & np.isnan(Y_g))))
print("differing Y entries", bad[:10].tolist(), "ystart", pa.ystart)
for cbounds in ([0, g.n], [0, 3, g.n], [0, 3, 4, g.n], [0, 4, g.n]):
    o2, X2, Y2, _ = cases.run_shards_on_one_device(_capi.hip_api(), pa, X0, Y0, q, [0, g.m] if len(cbounds) == 2 else np.linspace(0, g.m, len(cbounds)).astype(int).tolist(), cbounds)
    print("column bounds", cbounds, "objective", o2[:4], "equal to single:", np.array_equal(o2, o_g[1:]), np.array_equal(X2, X_g), np.array_equal(Y2, Y_g))
