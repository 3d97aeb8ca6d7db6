"""GLRM_PROBLEM_ROWS_FROM_COLS beyond one hipCUB sort (1.5e9 entries): the C5 recipe's Omega handed over as its column view alone, the row view
derived on the device in row ranges (csrc/glrm_transpose.hip), against the handle built from both views of the same generator output.

    python tests/perf/bench_rows_from_cols.py [--rows 2500000] [--config C5] [--iters 2]

Prints one JSON line: seconds of each create, and whether `iters` outer iterations give the same objectives bit for bit (the row view's
order inside a row decides the summation order, a wrong entry decides the value).  Both views live in HBM (DEVICE_ARRAYS): what is timed
is the device-side derivation, not PCIe."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C5")
    ap.add_argument("--rows", type=int, default=2_500_000)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--seed", type=int, default=20260926)
    args = ap.parse_args()
    import torch
    from bench_legs import CONFIGS
    from lowrankmodels.jl_amd import _capi, synth
    from lowrankmodels.jl_amd.fit import ShardedFit
    cfg = CONFIGS[args.config]
    m, n, k, q, reg = args.rows, cfg["cols"], cfg["k"], cfg["q"], cfg["reg"]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    api = _capi.hip_api()
    w = synth.DeviceWorkload(m, n, k, q, seed=args.seed, value_model=cfg["value_model"], loss_mix=cfg["loss_mix"], rx=reg, ry=reg, device=device)
    stream = torch.cuda.current_stream().cuda_stream

    class P:
        stepsize, inner_iter_X, inner_iter_Y, min_stepsize = 1.0, 1, 1, 0.01

    def run(prob):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sf = ShardedFit(api, prob, [0, m], [0, n], device=device, stream=stream, opts=dict(profile=0))
        torch.cuda.synchronize()
        t_create = time.perf_counter() - t0
        X0, Y0 = w.init_factors(sf.ld)
        sf.dX.copy_(X0); sf.dY.copy_(Y0)
        del X0, Y0
        api.reset_stepsizes(sf.h, 1.0)
        objs = [sf.initial_objective()] + [sf.iteration(P) for _ in range(args.iters)]
        st = api.kernel_stats(sf.h)
        sf.close()
        del sf
        torch.cuda.empty_cache()
        return t_create, objs, st

    t_both, o_both, st_both = run(w.problem())
    p = lambda t: int(t.data_ptr())
    pc = _capi.ProblemArrays(m, n, k, None, None, None, p(w.colptr), p(w.rowidx), p(w.colvals), w.losses, w.rx, w.ry,
                             flags=_capi.PROBLEM_ROWS_FROM_COLS | _capi.PROBLEM_DEVICE_ARRAYS)
    t_cols, o_cols, st_cols = run(pc)
    print(json.dumps({"config": args.config, "rows": m, "cols": n, "k": k, "observations": int(w.nnz_cols),
                      "row_ranges": -(-int(w.nnz_cols) // int(os.environ.get("GLRM_HIP_TRANSPOSE_CHUNK", "1500000000"))),
                      "create_both_views_s": t_both, "create_column_view_only_s": t_cols,
                      "objectives_both": o_both, "objectives_column_view_only": o_cols, "same_bits": o_both == o_cols,
                      "same_families": st_both["tiled"] == st_cols["tiled"], "nnz_rows": int(st_cols["nnz_rows"])}), flush=True)


if __name__ == "__main__":
    main()
