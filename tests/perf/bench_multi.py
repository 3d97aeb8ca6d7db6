"""Time the general sweeps (multi-dimensional losses) on a categorical model: m rows x n columns with K levels each,
MultinomialLoss / BvSLoss / MultinomialOrdinalLoss columns, rank k, fully observed.  Prints ms per half-step and
observation-updates/s; with --cpu also times the oracle on the same model.
    python tests/perf/bench_multi.py --m 200000 --n 100 --K 5 --k 10 --iters 5 [--cpu]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lowrankmodels.jl_amd as L  # noqa: E402
from lowrankmodels.jl_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=200000)
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--K", type=int, default=5)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--mix", default="mnl", help="mnl | ordinal | mixed")
ap.add_argument("--cpu", action="store_true")
a = ap.parse_args()

rng = np.random.default_rng(0)
m, n, K, k = a.m, a.n, a.K, a.k
if a.mix == "mnl":
    losses = [L.MultinomialLoss(K) for _ in range(n)]
    rx, ry = L.QuadReg(0.1), L.QuadReg(0.1)
elif a.mix == "ordinal":
    losses = [L.MultinomialOrdinalLoss(K) if f % 2 else L.BvSLoss(K) for f in range(n)]
    rx, ry = L.lastentry1(L.QuadReg(0.1)), [L.MNLOrdinalReg(L.QuadReg(0.1)) if f % 2 else L.OrdinalReg(L.QuadReg(0.1)) for f in range(n)]
else:
    losses = [[L.MultinomialLoss(K), L.QuadLoss(), L.BvSLoss(K), L.LogisticLoss()][f % 4] for f in range(n)]
    rx, ry = L.QuadReg(0.1), L.QuadReg(0.1)
Z = rng.standard_normal((m, 3))
A = np.zeros((m, n))
for f, lo in enumerate(losses):
    z = Z @ rng.standard_normal(3)
    A[:, f] = np.clip(np.round((1 + lo.max) / 2 + z), 1, lo.max) if hasattr(lo, "max") else (z > 0 if lo.classification else z)
D = L.embedding_dim(losses)
X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, D))
t0 = time.time()
g = L.GLRM(A, losses, rx, ry, k, X=X0, Y=Y0)
pa = g.problem_arrays()
nnz = int(pa.rowptr[-1])
print(f"model: m={m} n={n} K={K} k={k} D={D} nnz={nnz:.3g} ({time.time() - t0:.1f}s to build)", flush=True)


def run(api, label, iters):
    h = api.create(pa, profile=1) if label == "hip" else api.create(pa)
    try:
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        api.set_factors(h, X, Y)
        api.reset_stepsizes(h, 1.0)
        api.step_x(h, 0.01); api.step_y(h, 0.01)  # warm-up iteration (large first-step line search)
        if label == "hip":
            api.synchronize(h)
            api.kernel_stats(h, reset=True)
        tx = ty = 0.0
        for _ in range(iters):
            t = time.time(); api.step_x(h, 0.01)
            if label == "hip": api.synchronize(h)
            tx += time.time() - t
            t = time.time(); api.step_y(h, 0.01)
            if label == "hip": api.synchronize(h)
            ty += time.time() - t
        st = api.kernel_stats(h)
        obj = api.sum(h, None, 0) if False else None
        print(f"{label}: X half-step {1e3 * tx / iters:.2f} ms, Y half-step {1e3 * ty / iters:.2f} ms, "
              f"{nnz * iters / (tx + ty):.3g} obs-updates/s; trials x/y per segment-iter "
              f"{st['trials_x'] / (m * iters):.2f}/{st['trials_y'] / (n * iters):.2f}", flush=True)
    finally:
        api.destroy(h)


run(_capi.hip_api(), "hip", a.iters)
if a.cpu:
    import oracle as O
    O.set_threads(O.usable_cores())
    run(O.oracle_api(), f"cpu({O.usable_cores()}thr)", max(1, a.iters // 3))
