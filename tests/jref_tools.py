"""The J_ref parity helpers (they lived in bench.py and a test imported the benchmark script for them): the committed
fixtures tests/golden/jref_<config>.json / .npz (tools/make_jref.py: the CPU oracle's run to its own stop on >= 1e8 observations of a
BASELINE recipe -- whole trajectory, 512 rows of X, 512 columns of Y, in the reference's summation order and in the engine's), the same
problem regenerated in HBM, and the comparison of an engine run with both oracle runs.  Test infrastructure: tests/test_gpu_jref.py asserts
on the result, bench.py's `to_ref_objective` leg prints it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_jref_fixture(config, seed, cfg):
    """tests/golden/jref_<config>.json (+ .npz, tools/make_jref.py): the CPU oracle's run to its own stop on >= 1e8 observations of the
    recipe -- the whole objective trajectory and factor samples, in the reference's summation order and in the engine's."""
    path = os.path.join(ROOT, "tests", "golden", f"jref_{config}.json")
    if not os.path.exists(path):
        return None
    fx = json.load(open(path))
    c = cfg
    ok = fx["seed"] == seed and fx["k"] == c["k"] and fx["value_model"] == c["value_model"] and fx["loss_mix"] == c["loss_mix"] and tuple(fx["reg"]) == tuple(c["reg"])
    if not ok:
        return None
    npz = path[:-5] + ".npz"
    if os.path.exists(npz):
        import numpy as np
        fx["_samples"] = dict(np.load(npz))
    return fx


def jref_device_problem(fixture, cfg, seed, api, device, **create_kw):
    """The fixture's problem regenerated in HBM from the same counter-based generator (Omega and values are bit-identical between the
    device and the CPU generator, tests/test_synth.py): (handle, X0, Y0 as host k x m / k x n arrays)."""
    import numpy as np
    from lowrankmodels.jl_amd import synth
    ms, n, q, k = fixture["m"], fixture["n"], fixture["q"], cfg["k"]
    reg = cfg["reg"]
    w = synth.DeviceWorkload(ms, n, k, q, seed=seed, value_model=cfg["value_model"], loss_mix=cfg["loss_mix"], rx=reg, ry=reg, device=device)
    h = api.create(w.problem(), device_id=device.index or 0, **create_kw)
    w.free_sources()
    # the start comes from the CPU generator, like the fixture's: its Box-Muller normals go through the host's libm, the device
    # generator's through ocml -- the same numbers to the last bit or two, which is not the same start for a bit-for-bit comparison
    import oracle as O
    X0, Y0 = O.synth_cpu_init(ms, n, k, q, seed=seed)
    if cfg["reg"][0] == 3:  # NonNegConstraint configs start from |N(0,1)| / sqrt(k) (bench.py: INIT_NOTE)
        X0, Y0 = np.asfortranarray(np.abs(X0) * (1.0 / k ** 0.5)), np.asfortranarray(np.abs(Y0) * (1.0 / k ** 0.5))
    return h, X0, Y0


def trajectory_deviation(obj_gpu, obj_cpu):
    """max_i |obj_gpu[i] - obj_cpu[i]| / |obj_cpu[i]| over the recorded iterations both runs have (the initial objective included)."""
    import numpy as np
    nn = min(len(obj_gpu), len(obj_cpu))
    a, b = np.asarray(obj_gpu[:nn], dtype=np.float64), np.asarray(obj_cpu[:nn], dtype=np.float64)
    fin = np.isfinite(b)
    if not np.array_equal(fin, np.isfinite(a)):
        return {"iterations_compared": nn, "max_rel": float("inf"), "at_iteration": int(np.flatnonzero(fin != np.isfinite(a))[0])}
    d = np.zeros(nn)
    d[fin] = np.abs(a[fin] - b[fin]) / np.abs(b[fin])
    return {"iterations_compared": nn, "max_rel": float(d.max()) if nn else 0.0, "at_iteration": int(d.argmax()) if nn else 0,
            "rel_at_last": float(d[-1]) if nn else 0.0}


def jref_parity(fixture, api, h, X0, Y0):
    """north_star: "within 1e-5 relative on the objective trajectory and factor values".  The engine runs default ProxGradParams() -- its
    OWN stop rule -- on the fixture's problem and is compared, over ALL recorded iterations and on the stored factor samples, with
      engine_order     the oracle adding in the order the engine reports (glrm_hip_sum_order): must agree to the last bit of the factors
      reference_order  the oracle in the reference's order: what the north star's 1e-5 is about; the two oracle runs' own deviation
                       (summation order alone, stored in the fixture) is printed beside the engine's
    A second run with the stop rule off covers the reference-order run's length when the two orders stop at different iterations."""
    import numpy as np
    from lowrankmodels.jl_amd.params import ProxGradParams
    sm = fixture.get("_samples")
    eo = fixture.get("engine_order")
    out = {}
    orders = [api.sum_order(h, w).asdict() for w in (0, 1)]
    out["engine_sum_order"] = {"rows": orders[0], "cols": orders[1]}
    Xg, Yg = X0.copy(order="F"), Y0.copy(order="F")
    obj, sec = api.fit(h, ProxGradParams(), Xg, Yg)
    out["gpu_iterations_to_own_stop"] = len(obj) - 1
    out["gpu_seconds_to_own_stop"] = float(sec[-1])
    st = api.kernel_stats(h)
    out["line_search_totals"] = {k_: int(st[k_]) for k_ in ("trials_x", "trials_y", "accepts_x", "accepts_y")}

    def samples(tag, X, Y):
        r = {}
        for nm, F, ix in (("X", X, sm["rows"]), ("Y", Y, sm["cols"])):
            ref = sm[f"{nm}_{tag}"]
            got = F[:, ix]
            r[f"{nm}_sample_rel_fro"] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            r[f"{nm}_sample_bit_identical"] = bool(np.array_equal(got, ref))
        return r

    if eo:
        want = eo["orders"]
        same = all(orders[i].get(f) == v for i, side in enumerate(("rows", "cols")) for f, v in want[side].items())
        e = {"engine_reports_the_fixtures_order": bool(same), "cpu_iterations_to_own_stop": eo["iterations_to_own_stop"],
             "trajectory": trajectory_deviation(obj, eo["objective"]),
             "line_search_totals_equal": all(int(st[k_]) == int(eo["line_search"][k_]) for k_ in ("trials_x", "trials_y", "accepts_x", "accepts_y")) if len(obj) == len(eo["objective"]) else None}
        if sm is not None and len(obj) == len(eo["objective"]):
            e.update(samples("eng", Xg, Yg))
        out["vs_oracle_in_engine_order"] = e
        out["oracle_reference_vs_engine_order"] = eo["deviation_from_reference_order"]
    it_ref = int(fixture["iterations_to_own_stop"])
    r = {"cpu_iterations_to_own_stop": it_ref}
    if len(obj) - 1 != it_ref:  # the reference-order run stopped elsewhere: the same start, stop rule off, exactly that many iterations
        Xg, Yg = X0.copy(order="F"), Y0.copy(order="F")
        obj, _ = api.fit(h, ProxGradParams(max_iter=it_ref, abs_tol=-1e300, rel_tol=-1e300), Xg, Yg)
        st = api.kernel_stats(h)
    r["trajectory"] = trajectory_deviation(obj, fixture["objective"])
    if sm is not None:
        r.update(samples("ref", Xg, Yg))
    if "line_search" in fixture:
        tot = fixture["line_search"]
        r["line_search_agreement"] = {k_: {"gpu": int(st[k_]), "cpu": int(tot[k_])} for k_ in ("trials_x", "trials_y", "accepts_x", "accepts_y")}
    out["vs_oracle_in_reference_order"] = r
    out["tolerance"] = "north star: 1e-5 relative on trajectory and factor values; see DESIGN.md section 3 for what summation order alone does to it"
    return out


def jref_reference_mode(fixture, api, h_ref, X0, Y0):
    """The engine in its REFERENCE-ORDER validation mode (glrm_options.sum_order = 1, csrc/glrm_reforder.hip: one lane per segment, list
    order, one accumulator per sum, Julia's pairwise column sums) against the fixture's reference-order run -- the literal form of the
    north star's "within 1e-5 on the objective trajectory and factor values" on a recipe whose trajectory amplifies summation order
    (until round 5 that clause was attributed through the oracle's order switch, never met by an engine run)."""
    import numpy as np
    from lowrankmodels.jl_amd.params import ProxGradParams
    sm = fixture.get("_samples")
    orders = [api.sum_order(h_ref, w).asdict() for w in (0, 1)]
    Xg, Yg = X0.copy(order="F"), Y0.copy(order="F")
    obj, sec = api.fit(h_ref, ProxGradParams(), Xg, Yg)
    st = api.kernel_stats(h_ref)
    r = {"engine_reports": [o["family_name"] for o in orders], "kernel_flags": int(st["tiled"]),
         "gpu_iterations_to_own_stop": len(obj) - 1, "cpu_iterations_to_own_stop": int(fixture["iterations_to_own_stop"]),
         "gpu_seconds_to_own_stop": float(sec[-1]),
         "trajectory": trajectory_deviation(obj, fixture["objective"]),
         "trajectory_after_the_initial_objective": trajectory_deviation(obj[1:], fixture["objective"][1:]),
         # the whole recorded vector, objective[0] included (one accumulator over all observations, src/evaluate_fit.jl:12-21)
         "objective_vector_bit_identical": bool(np.array_equal(np.asarray(obj, dtype=np.float64), np.asarray(fixture["objective"], dtype=np.float64)))}
    if sm is not None and len(obj) == len(fixture["objective"]):
        for nm, F, ix in (("X", Xg, sm["rows"]), ("Y", Yg, sm["cols"])):
            ref = sm[f"{nm}_ref"]
            got = F[:, ix]
            r[f"{nm}_sample_rel_fro"] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            r[f"{nm}_sample_bit_identical"] = bool(np.array_equal(got, ref))
    if "line_search" in fixture:
        r["line_search_totals_equal"] = all(int(st[k_]) == int(fixture["line_search"][k_]) for k_ in ("trials_x", "trials_y", "accepts_x", "accepts_y"))
    return r
