#!/usr/bin/env python
"""J_ref fixtures: the CPU oracle's run to its own stop on >= 1e8 observations of a bench recipe, in TWO summation orders.

SURVEY.md 8(d) (iterations / wall-clock to the reference's own stop) and the north star's parity clause ("within 1e-5 relative on the
objective trajectory and factor values"): the oracle (oracle/, the restatement of src/algorithms/proxgrad.jl:34-220) runs default
ProxGradParams() -- its own stop rule, max_iter 100 -- on a scaled-down problem of the bench recipe that is still large enough for the
stop rule to mean something (1e8 observations: minutes to an hour of CPU time, so it is run ONCE, here, and committed).  bench.py and
tests/test_gpu_jref.py regenerate the identical problem on the GPU from the same counter-based generator.

Stored per config (tests/golden/jref_<config>.json + jref_<config>.npz):
  reference order   the whole objective trajectory, X[:, rows] and Y[:, cols] of 512 evenly spaced rows / columns after the last
                    iteration, trial / accept totals -- what the engine must match within 1e-5 (every iteration, factor samples)
  engine order      the same run with the oracle adding every segment's terms in the order the engine's kernels add them for THIS
                    problem (include/glrm_hip.h: glrm_sum_order; oracle/glrm_oracle.c: eng_pass, pinned to a lane-by-lane simulation of
                    the kernels by tests/test_sum_order.py).  The order is written down here from the engine's family rules
                    (expected_orders) and stored; the GPU test first asserts that glrm_hip_sum_order reports exactly it, then that the
                    engine's factor samples equal these BIT FOR BIT.  The deviation between the two stored runs is what summation order
                    alone does to this trajectory (the 8e-6 of round 3's bench line).

    python tools/make_jref.py C4 --rows 1000000          # -> tests/golden/jref_C4.json, jref_C4.npz
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

NSAMPLE = 512


def tile_rows(kp):
    return ((150 * 1024) // (kp * 8 + 16)) // 16 * 16  # csrc/glrm_tiled.hip: tile_rows_c(kp, 1)


def expected_orders(m, n, k, q, mixed=False):
    """The summation order the engine's auto choice lands on for a bench recipe problem of m x n, rank k, q sorted observations per row
    (csrc/glrm_tiled.hip: glrm_setup_tiled, csrc/glrm_cached.hip: glrm_setup_cached, csrc/glrm_blocked.hip: glrm_setup_blocked,
    csrc/glrm_hip.hip: glrm_hip_sum_order), as dicts in the field names of glrm_sum_order.  A statement of what is expected, checked
    against the engine's own report on the GPU (tests/test_gpu_jref.py) -- never a substitute for it."""
    kp = 8 if k <= 8 else 16 if k <= 16 else 32 if k <= 32 else 64 if k <= 64 else 128
    G = 4 if kp <= 32 else 8 if kp == 64 else 16
    R = kp // G
    nnz = m * q
    T = tile_rows(kp)
    base = dict(lanes=G, comps=R, waves=0, waves4_from=1536, waves8_from=98304, cached_maxlen=-1, cached_waves=0, batch=1, batch_one_wave_only=0,
                rotate=0, window=0, windows_per_sup=0, private_order=0)
    spb = 16 * (64 // G)
    tiled_r = q * T / n >= 4.0 and nnz >= 2e7 and m >= 512 * spb
    tiled_c = (nnz / n) * T / m >= 4.0 and nnz >= 2e7 and n >= 256
    rows, cols = dict(base), dict(base)
    # models with several loss kinds (C5): the whole batch of G observations per step, one loss partial per lane; the column passes walk
    # their chunks rotated; the row view is grouped by loss kind inside every window (private_order = 2, which the oracle restates)
    if mixed and not (tiled_r and tiled_c and G in (4, 8) and R == 8):
        raise SystemExit("a heterogeneous recipe outside the LDS-tiled families: write the gather families' batch rules down here first")
    # lane-per-segment form of the LDS-tiled passes (csrc/glrm_lane.hip: glrm_setup_lane): padded rank 32, at most 2e9 observations in the view;
    # rows of a model with a loss per column as well since session r6_33 (a descriptor per observation through one-byte ids: at most 256
    # distinct descriptors -- the bench recipes hold three).  Reported as the two-lane layout with the rotated chunk walk (rotate = 2); the row
    # view keeps the caller's order (no kind grouping)
    lane_r = tiled_r and kp == 32 and nnz <= 6_000_000_000
    lane_c = tiled_c and kp == 32 and nnz <= 6_000_000_000   # (beyond 2e9 observations on the compact form of the stream: same sums)
    if tiled_r:
        rows.update(family=2, window=T, windows_per_sup=0, batch=G if mixed else 2, private_order=2 if mixed else 0)
        if lane_r:
            rows.update(lanes=2, comps=kp // 2, batch=2, rotate=2, private_order=0)
    else:
        rows.update(family=1)
        if n * kp * 8 > 32 * 2 ** 20 and nnz >= 1e8 and G in (4, 8) and R == 8:
            rows.update(cached_maxlen=13 * (64 // G), cached_waves=2)
    if tiled_c:
        ntiles = -(-m // T)
        groups = -(-n // (512 if lane_c else spb))          # (glrm_setup_tiled: 512 columns per workgroup and twice the workgroups on the lane family)
        tps = max(1, min(ntiles // max(1, -(-(2048 if lane_c else 1024) // groups)), max(1, 32768 // T)))
        cols.update(family=2, window=T, windows_per_sup=tps, batch=G if mixed else 2, rotate=1 if mixed else 0)
        if lane_c:
            cols.update(lanes=2, comps=kp // 2, batch=2, rotate=2)
    else:
        if m * kp * 8 > 32 * 2 ** 20 and nnz >= 2e8:
            raise SystemExit("this problem would run the phase-aligned column passes: write their geometry down here first")
        cols.update(family=1)
    return rows, cols


def as_corder(d):
    from lowrankmodels.jl_amd import _capi
    o = _capi.CSumOrder()
    for f, v in d.items():
        setattr(o, f, v)
    return o


def run(api, pa, X0, Y0, orders):
    h = api.create(pa)
    if orders is not None:
        for w, d in enumerate(orders):
            O.set_sum_order(h, w, as_corder(d))
    X, Y = X0.copy(order="F"), Y0.copy(order="F")
    t0 = time.time()
    obj, _ = api.fit(h, ProxGradParams(), X, Y)
    t = time.time() - t0
    st = api.kernel_stats(h)
    api.destroy(h)
    return obj, X, Y, st, t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["C2", "C4", "C5"])
    ap.add_argument("--rows", type=int, required=True)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--obs-per-row", type=int, default=0)
    ap.add_argument("--seed", type=int, default=20260926)
    ap.add_argument("--out-tag", default="", help="suffix of the output files (cuts of the recipe)")
    ap.add_argument("--time-only", action="store_true", help="re-run the reference-order fit on THIS host's cores, check it against the committed fixture bit for "
                    "bit and print its seconds (the fixture is not rewritten): the CPU leg of `wall-clock to reference convergence` on the GPU box")
    a = ap.parse_args()
    cfg = dict(bench.CONFIGS[a.config])
    n, q, k = a.cols or cfg["cols"], a.obs_per_row or cfg["q"], cfg["k"]
    cores = O.usable_cores()
    O.set_threads(cores)
    t0 = time.time()
    pa, X0, Y0 = bench._oracle_problem(a.rows, n, k, q, cfg, a.seed)
    t_gen = time.time() - t0
    api = O.oracle_api()
    orders = expected_orders(a.rows, n, k, q, mixed=bool(cfg["loss_mix"]))
    path = os.path.join(ROOT, "tests", "golden", f"jref_{a.config}{a.out_tag}.json")
    old = json.load(open(path)) if os.path.exists(path) else None

    obj, X, Y, st, t_fit = run(api, pa, X0, Y0, None)
    print("reference order:", obj[-1], len(obj) - 1, f"{t_fit:.0f} s", flush=True)
    if a.time_only:
        same = old is not None and old.get("objective") == [float(v) for v in obj]
        print(json.dumps({"mode": "jref-time-only", "config": a.config, "m": a.rows, "n": n, "k": k, "observations": int(pa.rowptr[-1]), "cores": cores,
                          "iterations_to_own_stop": len(obj) - 1, "J_ref": float(obj[-1]), "cpu_seconds": t_fit, "cpu_seconds_per_iteration": t_fit / max(len(obj) - 1, 1),
                          "generate_seconds": t_gen, "trajectory_equals_the_committed_fixture_bit_for_bit": bool(same), "where": "this host"}), flush=True)
        return
    obj_e, X_e, Y_e, st_e, t_fit_e = run(api, pa, X0, Y0, orders)
    print("engine order:   ", obj_e[-1], len(obj_e) - 1, f"{t_fit_e:.0f} s", flush=True)

    rows = np.linspace(0, a.rows - 1, NSAMPLE).astype(np.int64)
    cols = np.linspace(0, n - 1, NSAMPLE).astype(np.int64)
    npz = os.path.join(ROOT, "tests", "golden", f"jref_{a.config}{a.out_tag}.npz")
    np.savez(npz, rows=rows, cols=cols, X_ref=X[:, rows], Y_ref=Y[:, cols], X_eng=X_e[:, rows], Y_eng=Y_e[:, cols])
    nn = min(len(obj), len(obj_e))
    dev = np.abs(obj_e[:nn] - obj[:nn]) / np.abs(obj[:nn])
    # keep the CPU timing of an undisturbed earlier run of the same (bit-identical) reference-order trajectory if there is one
    keep_time = old is not None and old.get("objective") == [float(v) for v in obj] and "cpu_seconds" in old
    out = {"config": a.config, "recipe": cfg["text"].format(m=a.rows, n=n, k=k, pct=100.0 * q / n), "m": a.rows, "n": n, "k": k, "q": q,
           "observations": int(pa.rowptr[-1]), "seed": a.seed, "value_model": cfg["value_model"], "loss_mix": cfg["loss_mix"], "reg": list(cfg["reg"]),
           "start": bench.INIT_NOTE[bench.nonneg_start(cfg)], "params": "ProxGradParams() defaults: stepsize 1, max_iter 100, abs_tol 1e-5, rel_tol 1e-4, min_stepsize 0.01",
           "J_ref": float(obj[-1]), "iterations_to_own_stop": len(obj) - 1, "objective": [float(v) for v in obj],
           "line_search": {key: int(st[key]) for key in ("trials_x", "trials_y", "accepts_x", "accepts_y")},
           "cpu_seconds": old["cpu_seconds"] if keep_time else t_fit, "cpu_seconds_per_iteration": (old["cpu_seconds"] if keep_time else t_fit) / max(len(obj) - 1, 1),
           "cpu_cores": old["cpu_cores"] if keep_time else cores,
           "cpu_where": "the build container (8 cores), not the GPU box", "generate_seconds": t_gen,
           "engine_order": {"orders": {"rows": orders[0], "cols": orders[1]},
                            "what": "the same oracle run adding every segment's terms in the order the engine's kernels add them for this problem "
                                    "(glrm_sum_order; tests/test_gpu_jref.py asserts the engine reports exactly these orders)",
                            "objective": [float(v) for v in obj_e], "iterations_to_own_stop": len(obj_e) - 1,
                            "line_search": {key: int(st_e[key]) for key in ("trials_x", "trials_y", "accepts_x", "accepts_y")},
                            "cpu_seconds": t_fit_e,
                            "deviation_from_reference_order": {"max_rel_over_trajectory": float(dev.max()), "at_iteration": int(dev.argmax()),
                                                               "rel_at_last_common_iteration": float(dev[-1]),
                                                               "X_sample_rel_fro": float(np.linalg.norm(X_e[:, rows] - X[:, rows]) / np.linalg.norm(X[:, rows])),
                                                               "Y_sample_rel_fro": float(np.linalg.norm(Y_e[:, cols] - Y[:, cols]) / np.linalg.norm(Y[:, cols]))}},
           "factor_samples": f"jref_{a.config}{a.out_tag}.npz: X[:, rows], Y[:, cols] of {NSAMPLE} evenly spaced rows / columns after the last iteration, both orders",
           "made_by": "tools/make_jref.py (oracle/libglrm_oracle.so, OpenMP over rows then columns)"}
    json.dump(out, open(path, "w"), indent=1)
    print(path, out["J_ref"], out["iterations_to_own_stop"], out["engine_order"]["deviation_from_reference_order"])


if __name__ == "__main__":
    main()
