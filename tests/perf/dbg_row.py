"""One row's line search on both engines at the point where they part.  python tests/perf/dbg_row.py SEED FULL_ITERS ROW"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, oracle as O
import importlib.util
from lowrankmodels.jl_amd import _capi
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
seed, full, row = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g, p = fz.random_model(seed)
pa = g.problem_arrays()
X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
names = [type(l).__name__ for l in g.losses]
out = {}
for label, api in (("cpu", O.oracle_api()), ("hip", _capi.hip_api())):
    h = api.create(pa)
    try:
        api.set_factors(h, X0, Y0); api.reset_stepsizes(h, p.stepsize)
        hist = []
        for it in range(full):
            s0 = dict(api.kernel_stats(h)); api.step_x_range(h, row, row + 1, p.min_stepsize); s1 = dict(api.kernel_stats(h))
            hist.append((s1["trials_x"] - s0["trials_x"], s1["accepts_x"] - s0["accepts_x"]))
            api.step_x_range(h, 0, row, p.min_stepsize); api.step_x_range(h, row + 1, pa.m, p.min_stepsize)
            api.step_y(h, p.min_stepsize)
        X, Y = np.zeros_like(X0), np.zeros_like(Y0); api.get_factors(h, X, Y)
        s0 = dict(api.kernel_stats(h)); api.step_x_range(h, row, row + 1, p.min_stepsize); s1 = dict(api.kernel_stats(h))
        X2, Y2 = np.zeros_like(X0), np.zeros_like(Y0); api.get_factors(h, X2, Y2)
        out[label] = (X, Y, X2, hist, (s1["trials_x"] - s0["trials_x"], s1["accepts_x"] - s0["accepts_x"]))
    finally:
        api.destroy(h)
c, hh = out["cpu"], out["hip"]
print("row", row, "columns", [(int(j), names[j]) for j in pa.colidx[pa.rowptr[row]:pa.rowptr[row + 1]]])
print("trials/accepts of the row in the earlier iterations: cpu", c[3], "hip", hh[3])
print("before: X fro", cases.fro_err(hh[0], c[0]), "Y fro", cases.fro_err(hh[1], c[1]), "row rel", np.max(np.abs(hh[0][:, row] - c[0][:, row]) / (np.abs(c[0][:, row]) + 1e-300)))
print("the row's step: cpu trials/accepts", c[4], "hip", hh[4])
print("x before (cpu)", c[0][:, row]); print("x after  (cpu)", c[2][:, row]); print("x after  (hip)", hh[2][:, row])
u = c[0][:, row] @ c[1]
ys = pa.ystart
for j in pa.colidx[pa.rowptr[row]:pa.rowptr[row + 1]]:
    print(f"   column {int(j)} {names[j]}: u = {u[ys[j]:ys[j + 1]]}, |y| = {np.abs(c[1][:, ys[j]:ys[j + 1]]).max():.3g}")
