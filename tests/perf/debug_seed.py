"""Where does the engine leave the oracle on a fuzz seed?  python tests/perf/debug_seed.py SEED [SEED ...]
Prints the model (shape, loss / regularizer kinds), then iteration by iteration the objective and the largest deviation of a row of X / a column
block of Y, and for the first deviating iteration the segments that deviate."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
import oracle as O  # noqa: E402
import lowrankmodels.jl_amd as L  # noqa: E402
from lowrankmodels.jl_amd import _capi  # noqa: E402

spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)


def main():
    O.set_threads(4)
    for seed in [int(a) for a in sys.argv[1:]]:
        g, p = fz.random_model(seed)
        pa = g.problem_arrays()
        X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
        stable = fz.well_conditioned_prefix(pa, X0, Y0, p, seed)
        kinds = sorted(set(int(l["kind"]) for l in pa.losses))
        print(f"seed {seed}: {g.m} x {g.n}, k = {g.k}, d = {pa.d}, nnz = {int(pa.rowptr[-1])}, loss kinds {kinds}, rx {sorted(set((int(r['kind']), int(r['wrap'])) for r in pa.rx))}, "
              f"ry {sorted(set((int(r['kind']), int(r['wrap'])) for r in pa.ry))}, stepsize {p.stepsize}, inner {p.inner_iter_X}/{p.inner_iter_Y}, stable prefix {stable}, offset {getattr(g, 'offset', False)}")
        ys = pa.ystart
        prev = None
        for it in range(1, stable + 1):
            q = L.ProxGradParams(p.stepsize, max_iter=it, inner_iter=p.inner_iter_X, abs_tol=0.0, rel_tol=-1.0)
            o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, q)
            o_g, X_g, Y_g, st_g = cases.run_engine(_capi.hip_api(), pa, X0, Y0, q)
            dx = np.abs(X_g - X_c).max(axis=0)
            dy = np.abs(Y_g - Y_c).max(axis=0)
            print(f"  it {it}: obj cpu {o_c[-1]!r} gpu {o_g[-1]!r}; max |dX| {dx.max():.3e} (row {int(dx.argmax())}), max |dY| {dy.max():.3e} (vector {int(dy.argmax())}); "
                  f"trials x {st_g['trials_x']}/{st_c['trials_x']} y {st_g['trials_y']}/{st_c['trials_y']} accepts y {st_g['accepts_y']}/{st_c['accepts_y']} flags {st_g['tiled']}")
            if max(dx.max(), dy.max()) > 1e-9 and prev is None:
                prev = it
                bad = np.flatnonzero(dy > 1e-9)
                for v in bad[:6]:
                    f = int(np.searchsorted(ys, v, side="right") - 1)
                    lo = pa.losses[f if len(pa.losses) > 1 else 0]
                    ry = pa.ry[f if len(pa.ry) > 1 else 0]
                    cnt = int(pa.colptr[f + 1] - pa.colptr[f])
                    print(f"     Y vector {int(v)} = column {f} (dims {int(ys[f])}..{int(ys[f + 1])}), loss {tuple(lo)}, ry {tuple(ry)}, {cnt} observations: cpu {Y_c[:, v][:4]} gpu {Y_g[:, v][:4]}")
                badx = np.flatnonzero(dx > 1e-9)
                for e in badx[:4]:
                    print(f"     X row {int(e)}: {int(pa.rowptr[e + 1] - pa.rowptr[e])} observations, rx {tuple(pa.rx[e if len(pa.rx) > 1 else 0])}: cpu {X_c[:, e][:4]} gpu {X_g[:, e][:4]}")


if __name__ == "__main__":
    main()
