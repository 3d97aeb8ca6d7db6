"""Soak of the lane-per-segment passes' forms (csrc/glrm_lane.hpp / .hip): random ragged problems at rank 32 forced onto the LDS tiles, the same
three iterations under every form of the trial rounds and both forms of the stream -- all of them must leave the same bits and the same
line-search counts.  Engine against engine: what is checked is that WHICH form ran changes nothing (the forms' sums against the oracle:
tests/test_gpu_sum_order.py).

    python tests/perf/soak_lane_forms.py FIRST_SEED LAST_SEED
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import lowrankmodels.jl_amd as L  # noqa: E402
from lowrankmodels.jl_amd import _capi  # noqa: E402

FORMS = {
    "default": {},
    "older forms": {"GLRM_HIP_LANE_ROUNDS": "0"},
    "chunk lists": {"GLRM_HIP_LANE_GATHER_TO": "101", "GLRM_HIP_LANE_GATHER_PACKED": "0"},
    "packed lists": {"GLRM_HIP_LANE_GATHER_TO": "101", "GLRM_HIP_LANE_GATHER_PACKED": "101", "GLRM_HIP_LANE_GATHER_SPREAD": "500"},
    "compact stream": {"GLRM_HIP_LANE_COMPACT": "1"},
    "compact stream, chunk lists": {"GLRM_HIP_LANE_COMPACT": "1", "GLRM_HIP_LANE_GATHER_TO": "101", "GLRM_HIP_LANE_GATHER_PACKED": "0"},
    "slots not dealt": {"GLRM_HIP_LANE_DEAL": "0"},
    "a wave per row": {"GLRM_HIP_LANE_TAIL": "101", "GLRM_HIP_LANE_TAIL_COLS": "101"},
    "no tail kernel": {"GLRM_HIP_LANE_TAIL": "0", "GLRM_HIP_LANE_TAIL_COLS": "0"},
}
KEYS = sorted({k for env in FORMS.values() for k in env})


def problem(seed):
    rng = np.random.default_rng(seed)
    if os.environ.get("SOAK_TINY"):   # shapes around one tile / one wave block / one chunk: shards without rows, single super-tiles
        m = int(rng.integers(40, 1500))
        n = int(rng.integers(40, 800))
    else:
        m = int(rng.integers(300, 40000))
        n = int(rng.integers(200, 5000))
    k = 32 if rng.random() < 0.8 else int(rng.integers(17, 33))   # padded rank 32 either way
    if os.environ.get("SOAK_RANK"):
        k = int(os.environ["SOAK_RANK"])
    skew = rng.random() < 0.5
    dens = rng.uniform(0.004, 0.06)
    if skew:   # power-law row degrees and column popularities
        wr = (np.arange(m) + 1.0) ** -rng.uniform(0.2, 0.8); rng.shuffle(wr)
        wc = (np.arange(n) + 1.0) ** -rng.uniform(0.2, 0.8); rng.shuffle(wc)
        p = np.minimum(1.0, dens * m * n * np.outer(wr / wr.sum(), wc / wc.sum()))
        mask = rng.random((m, n)) < p
    else:
        mask = rng.random((m, n)) < dens
    if rng.random() < 0.3:
        mask[rng.integers(0, m, size=max(1, m // 50))] = False     # empty rows
    mixed = rng.random() < 0.5
    kinds = [L.QuadLoss(0.8).descriptor(), L.HuberLoss(1.1, crossover=0.7).descriptor(), L.OrdinalHingeLoss(1, 5, 0.9).descriptor()]
    A = np.where(mask, np.round(rng.uniform(1, 5, size=(m, n))), 0.0)
    R = sp.csr_matrix((A[mask], np.nonzero(mask)), shape=(m, n))
    Cc = R.tocsc()
    Cc.sort_indices(); R.sort_indices()
    losses = np.array([kinds[f % 3] for f in range(n)] if mixed else [(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, float(rng.uniform(0.1, 2.0)))], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, R.indptr.astype(np.int64), R.indices.astype(np.int32), R.data.astype(np.float64),
                             Cc.indptr.astype(np.int64), Cc.indices.astype(np.int32), Cc.data.astype(np.float64), losses, reg, reg)
    X0 = np.asfortranarray(0.3 * rng.standard_normal((k, m)))
    Y0 = np.asfortranarray(0.3 * rng.standard_normal((k, n)))
    return pa, X0, Y0, dict(m=m, n=n, k=k, nnz=int(mask.sum()), skew=skew, mixed=mixed)


def run(api, pa, X0, Y0, env, start):
    for key in KEYS:
        os.environ.pop(key, None)
    os.environ.update(env)
    h = api.create(pa, tiled=2)
    try:
        flags = api.kernel_stats(h)["tiled"]
        api.set_factors(h, X0, Y0)
        api.reset_stepsizes(h, start)
        for _ in range(3):
            api.step_x(h, 0.01)
            api.step_y(h, 0.01)
        X, Y = np.zeros_like(X0), np.zeros_like(Y0)
        api.get_factors(h, X, Y)
        st = api.kernel_stats(h)
        return X, Y, {key: st[key] for key in ("trials_x", "trials_y", "accepts_x", "accepts_y")}, flags
    finally:
        api.destroy(h)


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    api = _capi.hip_api()
    bad = lane = 0
    for seed in range(first, last):
        pa, X0, Y0, info = problem(seed)
        start = [1.0, 64.0, 4096.0][seed % 3]
        ref = None
        for name, env in FORMS.items():
            X, Y, st, flags = run(api, pa, X0, Y0, env, start)
            if ref is None:
                ref = (X, Y, st, flags)
                lane += bool(flags & (256 | 512))
                continue
            same = np.array_equal(X, ref[0]) and np.array_equal(Y, ref[1]) and st == ref[2] and (flags == ref[3])
            if not same:
                bad += 1
                print("FAIL seed", seed, name, info, "flags", flags, ref[3], st, ref[2], "dX", float(np.abs(X - ref[0]).max()), "dY", float(np.abs(Y - ref[1]).max()), flush=True)
        if (seed - first) % 20 == 19:
            print("... seed", seed, info, "flags", ref[3], ref[2], flush=True)
    print(f"seeds {first}..{last - 1}: {last - first} problems ({lane} on the lane family), {len(FORMS) - 1} forms against the default each: {bad} differences")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
