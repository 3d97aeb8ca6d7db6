"""Per-iteration objective agreement (HIP vs oracle) of the fuzz models: python tests/perf/dbg_fuzz.py 1 16 17 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, oracle as O
import importlib.util
from lowrankmodels.jl_amd import _capi
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
for seed in map(int, sys.argv[1:]):
    g, p = fz.random_model(seed)
    pa = g.problem_arrays()
    X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, p)
    o_g, X_g, Y_g, st_g = cases.run_engine(_capi.hip_api(), pa, X0, Y0, p)
    with np.errstate(all="ignore"):
        rel = np.abs(o_g - o_c) / np.abs(o_c)
    kinds = sorted({type(l).__name__ for l in g.losses}); regs = sorted({repr(r) for r in list(g.rx)[:3] + list(g.ry)})
    print(seed, "m,n,k,d", g.m, g.n, g.k, g.d, "step", p.stepsize, "inner", p.inner_iter_X, "tiled", st_g["tiled"], "trials", st_c["trials_x"], st_g["trials_x"], st_c["trials_y"], st_g["trials_y"])
    print("   obj", " ".join("%.3g" % v for v in o_c[:8]))
    print("   rel", " ".join("%.1e" % v for v in rel))
    print("   X err %.2e Y err %.2e" % (cases.fro_err(X_g, X_c), cases.fro_err(Y_g, Y_c)), kinds, regs[:6])
