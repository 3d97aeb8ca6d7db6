"""Regenerate tests/golden/*.npz: inputs of the four SURVEY.md section 8(c) cases plus the CPU oracle's
trajectory (objective per iteration, final X and Y, line-search trial / accept totals).

The reference is pure Julia and cannot run in the build container (no julia binary), so these vectors
come from the oracle, whose operators are pinned to the reference's known answers by
tests/test_oracle_kat.py.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

import cases  # noqa: E402
import lowrankmodels.jl_amd as L  # noqa: E402
import oracle as O  # noqa: E402

if __name__ == "__main__":
    O.set_threads(1)
    api = O.oracle_api()
    for name in cases.GOLDEN_CASES:
        kwargs, params = cases.build_golden_case(name)
        pa = L.GLRM(**kwargs).problem_arrays()
        obj, X, Y, st = cases.run_engine(api, pa, kwargs["X"], kwargs["Y"], params)
        out = dict(objective=obj, X=X, Y=Y, trials=np.array([st["trials_x"], st["trials_y"]]),
                   accepts=np.array([st["accepts_x"], st["accepts_y"]]))
        path = os.path.join(HERE, name + ".npz")
        cases.save_case(path, kwargs, params, out)
        print(f"{name}: {len(obj) - 1} iterations, objective {obj[0]:.6g} -> {obj[-1]:.6g}, "
              f"trials {st['trials_x']}/{st['trials_y']}, accepts {st['accepts_x']}/{st['accepts_y']}, "
              f"{os.path.getsize(path) / 1024:.0f} kB")
