#!/usr/bin/env python
"""J_ref fixtures for bench.py's `to_ref_objective` leg (iterations / wall-clock to the reference's own stop).

The CPU oracle (oracle/, the restatement of src/algorithms/proxgrad.jl:34-220) runs default ProxGradParams() -- its own stop rule,
max_iter 100 -- on a scaled-down problem of the bench recipe that is still large enough for the stop rule to mean something (1e8
observations: minutes of CPU time, so it is run ONCE, here, and committed; bench.py regenerates the same problem on the GPU from the
same counter-based generator and reports the first iteration at or below J_ref (1 + 1e-5)).

    python tools/make_jref.py C4 --rows 1000000          # -> tests/golden/jref_C4.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import bench  # noqa: E402
import oracle as O  # noqa: E402
from lowrankmodels.jl_amd.params import ProxGradParams  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["C2", "C4", "C5"])
    ap.add_argument("--rows", type=int, required=True)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--obs-per-row", type=int, default=0)
    ap.add_argument("--seed", type=int, default=20260926)
    a = ap.parse_args()
    cfg = dict(bench.CONFIGS[a.config])
    n, q, k = a.cols or cfg["cols"], a.obs_per_row or cfg["q"], cfg["k"]
    cores = O.usable_cores()
    O.set_threads(cores)
    t0 = time.time()
    pa, X0, Y0 = bench._oracle_problem(a.rows, n, k, q, cfg, a.seed)
    t_gen = time.time() - t0
    api = O.oracle_api()
    h = api.create(pa)
    prm = ProxGradParams()
    X, Y = X0.copy(order="F"), Y0.copy(order="F")
    t0 = time.time()
    obj, sec = api.fit(h, prm, X, Y)
    t_fit = time.time() - t0
    api.destroy(h)
    out = {"config": a.config, "recipe": cfg["text"].format(m=a.rows, n=n, k=k, pct=100.0 * q / n), "m": a.rows, "n": n, "k": k, "q": q,
           "observations": int(pa.rowptr[-1]), "seed": a.seed, "value_model": cfg["value_model"], "loss_mix": cfg["loss_mix"], "reg": list(cfg["reg"]),
           "start": bench.INIT_NOTE[bench.nonneg_start(cfg)], "params": "ProxGradParams() defaults: stepsize 1, max_iter 100, abs_tol 1e-5, rel_tol 1e-4, min_stepsize 0.01",
           "J_ref": float(obj[-1]), "iterations_to_own_stop": len(obj) - 1, "objective": [float(v) for v in obj],
           "cpu_seconds": t_fit, "cpu_seconds_per_iteration": t_fit / max(len(obj) - 1, 1), "cpu_cores": cores,
           "cpu_where": "the build container (8 cores), not the GPU box", "generate_seconds": t_gen,
           "made_by": "tools/make_jref.py (oracle/libglrm_oracle.so, OpenMP over rows then columns)"}
    path = os.path.join(ROOT, "tests", "golden", f"jref_{a.config}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, out["J_ref"], out["iterations_to_own_stop"], t_fit)


if __name__ == "__main__":
    main()
