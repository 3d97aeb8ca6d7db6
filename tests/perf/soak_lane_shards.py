"""Soak of the lane-per-segment passes under sharding: the random ragged problems of soak_lane_forms.py, a four-iteration fit on ONE handle
against the in-library host with 2 .. 8 shards of the same device (row blocks swept in x_chunks sub-ranges: glrm_hip_step_x_range, gathered
rounds on sub-ranges; column blocks with their own class offsets) -- the same bits for every shard count.

    python tests/perf/soak_lane_shards.py FIRST_SEED LAST_SEED
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))

import numpy as np  # noqa: E402

import lowrankmodels.jl_amd as L  # noqa: E402
from lowrankmodels.jl_amd import _capi  # noqa: E402
from soak_lane_forms import problem  # noqa: E402


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    tiled = int(os.environ.get("SOAK_TILED", "2"))   # 2: LDS tiles (the lane family at rank 32); 1: gather sweeps; 0: the engine's own choice
    api = _capi.hip_api()
    bad = 0
    for seed in range(first, last):
        pa, X0, Y0, info = problem(seed)
        rng = np.random.default_rng(10_000_000 + seed)
        prm = L.ProxGradParams(stepsize=[1.0, 64.0, 4096.0][seed % 3], max_iter=4, abs_tol=-1e300, rel_tol=-1e300)
        h = api.create(pa, tiled=tiled)
        try:
            flags = api.kernel_stats(h)["tiled"]
            X1, Y1 = X0.copy(order="F"), Y0.copy(order="F")
            o1, _ = api.fit(h, prm, X1, Y1)
        finally:
            api.destroy(h)
        for n in sorted(set(int(v) for v in rng.integers(2, 9, size=2))):
            chunks = int(rng.integers(0, 5))
            mh = api.multi_create(pa, n, device_ids=[0] * n, x_chunks=chunks, tiled=tiled, arrival=int(rng.integers(0, 3)))
            try:
                X, Y = X0.copy(order="F"), Y0.copy(order="F")
                o, _ = api.multi_fit(mh, prm, X, Y)
            finally:
                api.multi_destroy(mh)
            same = np.array_equal(X, X1) and np.array_equal(Y, Y1) and len(o) == len(o1) and np.array_equal(o[1:], o1[1:])
            if not same:
                bad += 1
                print("FAIL seed", seed, "shards", n, "x_chunks", chunks, info, "flags", flags, "dX", float(np.abs(X - X1).max()), "dY", float(np.abs(Y - Y1).max()), flush=True)
        if (seed - first) % 20 == 19:
            print("... seed", seed, info, "flags", flags, flush=True)
    print(f"seeds {first}..{last - 1}: {last - first} problems, two shard counts each against one handle: {bad} differences")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
