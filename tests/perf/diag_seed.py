"""Drill-down for ONE seed of the randomized parity test (tests/test_gpu_fuzz.py::random_model): HIP engine against the oracle half-step by
half-step through the step-level API, printing where the two first differ by more than 1e-9 -- iteration, half-step, the worst rows /
columns with their loss kinds and regularizers.
    python tests/perf/diag_seed.py SEED [iterations]
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import lowrankmodels.jl_amd as L  # noqa: E402
from lowrankmodels.jl_amd import _capi  # noqa: E402

spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)


def main():
    seed = int(sys.argv[1])
    g, p = fz.random_model(seed)
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else p.max_iter
    pa = g.problem_arrays()
    X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
    ys = pa.ystart
    print(f"seed {seed}: m {g.m} n {g.n} k {g.k} d {pa.d} stepsize {p.stepsize} inner {p.inner_iter_X} offset {getattr(g, 'offset', None)}")
    O.set_threads(4)
    hs = []
    for api in (O.oracle_api(), _capi.hip_api()):
        h = api.create(pa)
        api.set_factors(h, X0, Y0)
        api.reset_stepsizes(h, p.stepsize)
        hs.append((api, h))
    print("engine families:", hs[1][0].kernel_stats(hs[1][1])["tiled"])

    def factors(api, h):
        X, Y = np.zeros_like(X0), np.zeros_like(Y0)
        api.get_factors(h, X, Y)
        return X, Y

    for it in range(1, iters + 1):
        if p.inner_iter_X > 1 or p.inner_iter_Y > 1:  # src/algorithms/proxgrad.jl:112-115: the step sizes start over in every outer iteration
            for api, h in hs:
                api.reset_stepsizes(h, p.stepsize)
        for half in ("x", "y"):
            for inner in range(p.inner_iter_X if half == "x" else p.inner_iter_Y):
                for api, h in hs:
                    (api.step_x if half == "x" else api.step_y)(h, p.min_stepsize)
            (Xc, Yc), (Xg, Yg) = factors(*hs[0]), factors(*hs[1])
            F_c, F_g = (Xc, Xg) if half == "x" else (Yc, Yg)
            nb = np.linalg.norm(F_c, axis=0)
            e = np.linalg.norm(F_g - F_c, axis=0) / np.where(nb > 0, nb, 1.0)
            worst = np.argsort(e)[::-1][:4]
            flag = "  <-- " if e.max() > 1e-9 else ""
            print(f"iter {it} {half}: max vector deviation {e.max():.2e} at {worst.tolist()} {[f'{v:.1e}' for v in e[worst]]}{flag}")
            if e.max() > 1e-9 and half == "y":
                for v in worst[:2]:
                    f = int(np.searchsorted(ys, v, side="right") - 1)
                    print(f"      Y vector {v} belongs to column {f}: {type(g.losses[f]).__name__} {g.losses[f].descriptor()} ry {type(g.ry[f]).__name__ if isinstance(g.ry, list) else type(g.ry).__name__}, "
                          f"{int(pa.colptr[f + 1] - pa.colptr[f])} observations")
            if e.max() > 1e-9 and half == "x":
                for v in worst[:2]:
                    cols = pa.colidx[pa.rowptr[v]:pa.rowptr[v + 1]]
                    kinds = sorted({type(g.losses[int(c)]).__name__ for c in cols})
                    print(f"      row {v}: {len(cols)} observations, losses {kinds}, rx {type(g.rx[v]).__name__ if isinstance(g.rx, list) else type(g.rx).__name__}")
        sts = [api.kernel_stats(h) for api, h in hs]
        print(f"        trials x {sts[0]['trials_x']} / {sts[1]['trials_x']}  y {sts[0]['trials_y']} / {sts[1]['trials_y']}   accepts x {sts[0]['accepts_x']} / {sts[1]['accepts_x']}  y {sts[0]['accepts_y']} / {sts[1]['accepts_y']}")
    for api, h in hs:
        api.destroy(h)


if __name__ == "__main__":
    main()
