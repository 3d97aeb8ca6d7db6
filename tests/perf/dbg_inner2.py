"""After every inner step of the first iteration: HIP vs oracle, and the oracle vs its own 1e-13-perturbed runs, on one Y vector.
python tests/perf/dbg_inner2.py SEED VECTOR"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, oracle as O
import importlib.util
from lowrankmodels.jl_amd import _capi
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
seed, vec = int(sys.argv[1]), int(sys.argv[2])
g, p = fz.random_model(seed)
pa = g.problem_arrays()
X0, Y0 = np.asfortranarray(g.X), np.asfortranarray(g.Y)
inner = p.inner_iter_X
def run(api, Xs, Ys):
    h = api.create(pa); out = []
    try:
        api.set_factors(h, Xs, Ys); api.reset_stepsizes(h, p.stepsize)
        for it in range(2):
            for _ in range(inner): api.step_x(h, p.min_stepsize)
            for _ in range(inner):
                api.step_y(h, p.min_stepsize)
                X, Y = np.zeros_like(X0), np.zeros_like(Y0); api.get_factors(h, X, Y); out.append((X, Y, dict(api.kernel_stats(h))))
    finally:
        api.destroy(h)
    return out
base = run(O.oracle_api(), X0, Y0)
hip = run(_capi.hip_api(), X0, Y0)
rng = np.random.default_rng(5)
pert = [run(O.oracle_api(), np.asfortranarray(X0 * (1 + 1e-13 * rng.standard_normal(X0.shape))), np.asfortranarray(Y0 * (1 + 1e-13 * rng.standard_normal(Y0.shape)))) for _ in range(8)]
rel = lambda a, b: float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300)))
for i, (b, h) in enumerate(zip(base, hip)):
    pv = [rel(q[i][1][:, vec], b[1][:, vec]) for q in pert]
    print(f"after Y inner step {i + 1}: vector {vec}: hip vs cpu {rel(h[1][:, vec], b[1][:, vec]):.2e}; perturbed cpu vs cpu: max {max(pv):.2e} median {np.median(pv):.2e}; "
          f"Y fro hip {cases.fro_err(h[1], b[1]):.2e} perturbed max {max(cases.fro_err(q[i][1], b[1]) for q in pert):.2e}; trials_y hip {h[2]['trials_y']} cpu {b[2]['trials_y']} accepts_y {h[2]['accepts_y']} {b[2]['accepts_y']}; y = {b[1][:, vec][:3]}")
