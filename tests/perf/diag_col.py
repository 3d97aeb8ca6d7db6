"""Drill-down of soak seed 7365 (tests/perf/soak_fuzz.py): column 36 (MultinomialOrdinalLoss, NonNegConstraint) at outer iteration 4 as a
one-column problem -- the objective of the engine and of the oracle at the current point and at every trial point of the line search,
and what one Y half-step of each does with it."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
from lowrankmodels.jl_amd import _capi  # noqa: E402

spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
seed, f, it_stop = int(sys.argv[1]) if len(sys.argv) > 1 else 7365, int(sys.argv[2]) if len(sys.argv) > 2 else 36, int(sys.argv[3]) if len(sys.argv) > 3 else 4
g, p = fz.random_model(seed)
pa = g.problem_arrays()
X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
oapi = O.oracle_api()
O.set_threads(4)
h = oapi.create(pa)
oapi.set_factors(h, X0, Y0)
for it in range(1, it_stop + 1):
    oapi.reset_stepsizes(h, p.stepsize)
    for _ in range(p.inner_iter_X):
        oapi.step_x(h, p.min_stepsize)
    if it < it_stop:
        for _ in range(p.inner_iter_Y):
            oapi.step_y(h, p.min_stepsize)
X, Y = np.zeros_like(X0), np.zeros_like(Y0)
oapi.get_factors(h, X, Y)
oapi.destroy(h)
ys = pa.ystart
d = int(ys[f + 1] - ys[f])
rows = pa.rowidx[pa.colptr[f]:pa.colptr[f + 1]].astype(np.int32)
vals = pa.colvals[pa.colptr[f]:pa.colptr[f + 1]].copy()
m, k = pa.m, pa.k
order = np.argsort(rows, kind="stable")
rowptr = np.zeros(m + 1, np.int64)
np.add.at(rowptr, rows + 1, 1)
rowptr = np.cumsum(rowptr)
one = _capi.ProblemArrays(m, 1, k, rowptr, np.zeros(len(rows), np.int32), vals[order].copy(), np.array([0, len(rows)], np.int64), rows.copy(), vals,
                          pa.losses[f:f + 1].copy(), pa.rx[:1].copy() if len(pa.rx) == 1 else pa.rx.copy(), pa.ry[f:f + 1].copy() if len(pa.ry) > 1 else pa.ry.copy())
Yb = np.asfortranarray(Y[:, ys[f]:ys[f + 1]].copy())
print(f"column {f}: dim {d}, {len(rows)} observations, loss {pa.losses[f]}, ry {one.ry}")
apis = {"oracle": oapi, "engine": _capi.hip_api()}
hs = {n: a.create(one) for n, a in apis.items()}
J0 = {n: apis[n].objective(hs[n], X, Yb, include_reg=False) for n in apis}
print("J(y):", {n: repr(v) for n, v in J0.items()})
# the gradient from the oracle's side (finite check only needs the trial points): G = sum_i x_i grad_i'
Gm = np.zeros_like(Yb)
lo = g.losses[f]
for i, a in zip(rows, vals):
    Gm += np.outer(X[:, i], O.vloss_grad(lo, X[:, i] @ Yb, a))
alpha, l = p.stepsize, len(rows) + 1
while alpha > p.min_stepsize:
    Yn = np.asfortranarray(np.maximum(Yb - (alpha / l) * Gm, 0.0))
    Jn = {n: apis[n].objective(hs[n], X, Yn, include_reg=False) for n in apis}
    print(f"alpha {alpha:.6f}: " + "  ".join(f"{n}: J(y') - J(y) = {Jn[n] - J0[n]:+.3e}" for n in apis))
    alpha *= 0.7
for n, a in apis.items():
    a.set_factors(hs[n], X, Yb)
    a.reset_stepsizes(hs[n], p.stepsize)
    a.step_y(hs[n], p.min_stepsize)
    Xo, Yo = np.zeros_like(X), np.zeros_like(Yb)
    a.get_factors(hs[n], Xo, Yo)
    st = a.kernel_stats(hs[n])
    print(f"{n}: one Y half-step: trials {st['trials_y']} accepts {st['accepts_y']} block changed {not np.array_equal(Yo, Yb)} by {np.linalg.norm(Yo - Yb):.3e}")
    a.destroy(hs[n])

# which observation's loss differs between y and y' (alpha = 0.020177) on the engine?  One-observation problems.
alpha = p.stepsize * 0.7 ** 9
Yn = np.asfortranarray(np.maximum(Yb - (alpha / l) * Gm, 0.0))
print("vectors of the block that moved:", [j for j in range(d) if not np.array_equal(Yn[:, j], Yb[:, j])])
eng = _capi.hip_api()
for t in range(len(rows)):
    rp = np.zeros(m + 1, np.int64)
    rp[rows[t] + 1:] = 1
    single = _capi.ProblemArrays(m, 1, k, rp, np.zeros(1, np.int32), vals[t:t + 1].copy(), np.array([0, 1], np.int64), rows[t:t + 1].copy(), vals[t:t + 1].copy(),
                                 one.losses, one.rx, one.ry)
    out = []
    for a_ in (oapi, eng):
        hh = a_.create(single)
        out.append((a_.objective(hh, X, Yb, include_reg=False), a_.objective(hh, X, Yn, include_reg=False)))
        a_.destroy(hh)
    u, un = X[:, rows[t]] @ Yb, X[:, rows[t]] @ Yn
    flag = "  <-- engine differs" if out[1][0] != out[1][1] else ""
    if flag or out[0][0] != out[0][1]:
        print(f"obs {t}: level {vals[t]:.0f}  u = {np.array2string(u, precision=6)}  u' = {np.array2string(un, precision=6)}  oracle {out[0][0]!r} {out[0][1]!r}  engine {out[1][0]!r} {out[1][1]!r}{flag}")
print("(observations not listed: identical loss at y and y' on both)")
