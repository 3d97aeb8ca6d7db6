#!/usr/bin/env python
"""bench.py -- observed-entry updates/sec of the GLRM proximal-gradient hot path on MI355X.

A "step" is one outer iteration of fit!(glrm, ProxGradParams) (X half-step over every observed entry of the
rank's rows, Y half-step over every observed entry of its columns, objective) on BASELINE.json configs[1]:
1M x 10k, rank 32, QuadLoss, 5 % observed (5e8 observations), QuadReg on X and Y, fp64 -- generated in HBM by
the counter-based generator (synthetic).  N > 1: one process per GPU (torch.distributed, nccl = RCCL), weak
scaling in m (each rank owns 1M rows and n/N columns; X and Y replicated; all-gather of the updated factor
after each half-step).  One observed-entry update = one (i,j) consumed by one factor half-step, so an outer
iteration performs |Omega_rows| + |Omega_cols| updates.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def algorithmic_bytes_per_update(k):
    """P = 2 compulsory passes x (8 B value + 4 B index + k x 8 B factor slice)."""
    return 2 * (8 + 4 + 8 * k)


def cpu_baseline(args, k, q, n):
    """The oracle (CPU restatement of the reference, oracle/) timed on the host cores on a bounded sample of the
    same workload: the first `sample_rows` rows x all n columns of the same generator."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle as O
    from lowrankmodels.jl_amd import _capi
    cores = O.usable_cores()  # affinity mask and cgroup CPU quota, not the hardware thread count of the host
    # enough rows for every thread to have work: 10 000 rows per thread, at least --cpu-sample-rows, at most 400k rows
    # (2e8 observations, ~5 GB)
    ms = int(min(max(args.cpu_sample_rows, 10_000 * cores), 400_000, args.rows_per_gpu))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(ms, n, k, q, seed=args.seed)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(ms, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    api = O.oracle_api()
    O.set_threads(cores)
    h = api.create(pa)
    api.set_factors(h, X0, Y0)
    api.reset_stepsizes(h, 1.0)
    iters = 0
    api.step_x(h, 0.01); api.step_y(h, 0.01)  # warm-up iteration
    t0 = time.time()
    while iters < 3 or (time.time() - t0 < 10.0 and iters < 500):
        api.step_x(h, 0.01)
        api.step_y(h, 0.01)
        iters += 1
    dt = time.time() - t0
    api.destroy(h)
    ups = iters * (int(rowptr[-1]) + int(colptr[-1])) / dt
    return {"value": ups, "unit": "observed-entry updates/s", "cores": cores, "kind": "port",
            "sample": f"first {ms} rows x all {n} columns of the same generator ({int(rowptr[-1])} observations), "
                      f"{iters} outer iterations after 1 warm-up, OpenMP over rows then columns"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--cols", type=int, default=10_000)
    ap.add_argument("--rank", dest="k", type=int, default=32)
    ap.add_argument("--obs-per-row", type=int, default=500)
    ap.add_argument("--seed", type=int, default=20260926)
    ap.add_argument("--waves-row", type=int, default=0)
    ap.add_argument("--waves-col", type=int, default=0)
    ap.add_argument("--tiled", type=int, default=0, help="0 auto, 1 gather sweeps only, 2 LDS-tiled sweeps")
    ap.add_argument("--x-chunks", type=int, default=4, help="N > 1: row chunks of the X half-step whose all-gather overlaps the next chunk")
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"],
                    help="BASELINE.json config family: C2 (default, the bench line), C4 = rank-64 NNMF 0.1 %% observed, "
                         "C5 = mixed Quad/Logistic/OrdinalHinge columns 2 %% observed (use --rows-per-gpu to scale)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-convergence-run", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=20_000)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lowrankmodels.jl_amd import _capi, synth
    from lowrankmodels.jl_amd.fit import ShardedFit

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # GLRM_BENCH_BACKEND=gloo is a plumbing check of the multi-rank path on a box with fewer GPUs than ranks (ranks then share
    # devices); the measured configuration is one rank per GPU over RCCL.
    backend = os.environ.get("GLRM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    value_model, loss_mix, reg = 0, 0, (1, 0, 1.0)  # QuadReg(1.0)
    if args.config == "C4":    # 10M x 100k rank 64, 1e9 observations, NonNegConstraint on X and Y
        args.cols, args.k, args.obs_per_row, value_model, reg = 100_000, 64, 100, 1, (3, 0, 1.0)
    elif args.config == "C3":  # 1M x 10k rank 32, fully observed QuadLoss, ZeroReg: the dense MFMA path (single GPU)
        args.obs_per_row, reg = args.cols, (0, 0, 1.0)
        if world != 1:
            raise SystemExit("--config C3 runs on one GPU (the dense hand-over takes the whole matrix)")
    elif args.config == "C5":  # 5M x 50k rank 32, 2 % observed, Quad/Logistic/OrdinalHinge columns, QuadReg
        args.cols, args.k, args.obs_per_row, loss_mix = 50_000 - 50_000 % (1000 * 1), 32, 1000, 1
    k, q, n = args.k, args.obs_per_row, args.cols
    m = args.rows_per_gpu * world  # weak scaling in m
    if n % world or n % q:
        raise SystemExit("cols must be divisible by the number of GPUs and by obs-per-row")
    rbs = [args.rows_per_gpu * r for r in range(world + 1)]
    cbs = [n // world * r for r in range(world + 1)]

    api = _capi.hip_api()
    t_gen = time.time()
    if args.config == "C3":
        w = synth.DenseDeviceWorkload(m, n, k, seed=args.seed, rx=reg, ry=reg, device=device)
    else:
        w = synth.DeviceWorkload(m, n, k, q, rows=(rbs[rank], rbs[rank + 1]), cols=(cbs[rank], cbs[rank + 1]), seed=args.seed,
                                 value_model=value_model, loss_mix=loss_mix, rx=reg, ry=reg, device=device)
    t_gen = time.time() - t_gen
    t_create = time.time()
    sf = ShardedFit(api, w.problem(), rbs, cbs, device=device, stream=torch.cuda.current_stream().cuda_stream,
                    opts=dict(profile=1, waves_row=args.waves_row, waves_col=args.waves_col, tiled=args.tiled),
                    x_chunks=args.x_chunks if args.config != "C3" else 1)
    nnz_r, nnz_c = w.nnz_rows, w.nnz_cols
    w.free_sources()
    t_create = time.time() - t_create
    X0, Y0 = w.init_factors(sf.ld)
    sf.dX.copy_(X0); sf.dY.copy_(Y0)
    del X0, Y0
    api.reset_stepsizes(sf.h, 1.0)

    class P:  # reference defaults, stop rule disabled (fixed number of outer iterations)
        stepsize, inner_iter_X, inner_iter_Y, min_stepsize = 1.0, 1, 1, 0.01

    obj0 = sf.initial_objective()
    objs = []
    for _ in range(args.warmup):
        objs.append(sf.iteration(P))
    api.kernel_stats(sf.h, reset=True)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        objs.append(sf.iteration(P))
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([nnz_r, nnz_c], dtype=torch.int64, device=device)
        dist.all_reduce(tot)
        tot_r, tot_c = int(tot[0].item()), int(tot[1].item())
    else:
        tot_r, tot_c = nnz_r, nnz_c
    st = api.kernel_stats(sf.h)

    # Second leg of the metric (iters-to-ref-objective): the reference's own run -- default ProxGradParams(), stop rule
    # of src/algorithms/proxgrad.jl:210-213, from the same X0, Y0 -- timed end to end.  Parity makes the iteration count
    # the reference's iteration count; the objective reached is what the reference would record.
    conv = None
    if not args.no_convergence_run:
        X0, Y0 = w.init_factors(sf.ld)
        sf.dX.copy_(X0); sf.dY.copy_(Y0)
        del X0, Y0
        api.reset_stepsizes(sf.h, 1.0)
        fence()
        tc = time.perf_counter()
        hist = [sf.initial_objective()]
        scaled_abs_tol = 1e-5 * float(tot_r)
        for i in range(1, 101):
            obj = sf.iteration(P)
            dec = hist[-1] - obj
            hist.append(obj)
            if i > 10 and (dec < scaled_abs_tol or dec / obj < 1e-4):
                break
        fence()
        conv = {"iterations": len(hist) - 1, "seconds": time.perf_counter() - tc, "objective_initial": hist[0],
                "objective_final": hist[-1], "stop_rule": "reference defaults: abs_tol=1e-5*|Omega|, rel_tol=1e-4, max_iter=100"}

    if rank == 0:
        updates_per_step = tot_r + tot_c
        value = args.steps * updates_per_step / elapsed
        bpu = algorithmic_bytes_per_update(k)
        ms_x = st["ms_x"] / max(args.steps, 1)  # per outer iteration (the X half-step may run as several chunk launches)
        ms_y = st["ms_y"] / max(args.steps, 1)
        tiled_row, tiled_col = bool(st["tiled"] & 1), bool(st["tiled"] & 2)
        # dominant kernel = the longest single kernel.  The gather column sweep and both row sweeps are one kernel per
        # half-step; the LDS-tiled column sweep is four launches (pass, reduce, trial pass, decide) of which the two
        # passes are about half the half-step each, so there the row sweep kernel is the longest.
        dom = "row_sweep" if (ms_x >= ms_y or tiled_col) else "col_sweep"
        dom_ms, dom_nnz = (ms_y, nnz_c) if dom == "col_sweep" else (ms_x, nnz_r)
        dom_tiled = tiled_col if dom == "col_sweep" else tiled_row
        dom_name = {("row_sweep", True): "tiled_sweep_kernel (X half-step, LDS-tiled)", ("row_sweep", False): "sweep_kernel<WAVES=1> (X half-step, gather)",
                    ("col_sweep", False): "sweep_kernel<WAVES=4> (Y half-step, gather)", ("col_sweep", True): "tiled_col_pass_kernel x2 + reduce/decide (Y half-step)"}[(dom, dom_tiled)]
        achieved = dom_nnz * bpu / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{dom}_k{k}_{'tiled' if dom_tiled else 'gather'}_bytes_per_launch")
            except Exception:
                traffic = None
        nseg_r, nseg_c = rbs[1] - rbs[0], cbs[1] - cbs[0]
        out = {
            "metric": "observed-entry updates/sec", "value": value, "unit": "updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[1] (C2): {m} x {n}, rank {k}, QuadLoss, {100.0 * q / n:.3g}% observed "
                                    f"({tot_r} observations), QuadReg(1.0) on X and Y, ProxGradParams defaults, stop rule off")
                       if args.config == "C2" else f"{args.config}-family: {m} x {n}, rank {k}, {tot_r} observations, "
                       f"{'NonNegConstraint' if reg[0] == 3 else 'QuadReg(1.0)'} on X and Y, {'mixed Quad/Logistic/OrdinalHinge' if loss_mix else 'QuadLoss'}",
                       "m": m, "n": n, "k": k, "observed": tot_r, "parallelism": f"rows/cols sharded over {world} GPU(s), X,Y replicated",
                       "waves_row": st["waves_row"], "waves_col": st["waves_col"],
                       "row_sweep": "lds-tiled" if tiled_row else "gather", "col_sweep": "lds-tiled" if tiled_col else "gather"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_update": bpu, "updates_per_launch": dom_nnz, "avg_launch_ms": dom_ms,
                         "traffic_GBps": traffic / (dom_ms * 1e-3) / 1e9 if traffic and dom_ms > 0 else None,
                         "traffic_frac": traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic and dom_ms > 0 else None,
                         "note": "algorithmic bytes assume every update fetches its own k-vector (SURVEY.md 8(d)); the LDS-tiled "
                                 "sweeps fetch a factor vector once per workgroup and tile, so frac > 1 means on-chip reuse, "
                                 "and `traffic` (PMC) is what actually crossed the fabric "
                                 "(traffic_frac = that over the measured launch time over the HBM peak; the tiled sweeps are "
                                 "LDS / VALU co-limited, DESIGN.md 4.2)"},
            "kernels": {"row_sweep_ms": ms_x, "col_sweep_ms": ms_y,
                        "row_sweep_GBps_algorithmic": nnz_r * bpu / (ms_x * 1e-3) / 1e9 if ms_x > 0 else None,
                        "col_sweep_GBps_algorithmic": nnz_c * bpu / (ms_y * 1e-3) / 1e9 if ms_y > 0 else None,
                        "mean_trials_per_row": st["trials_x"] / max(args.steps * nseg_r, 1),
                        "mean_trials_per_col": st["trials_y"] / max(args.steps * nseg_c, 1)},
            "objective": {"initial": obj0, "after_warmup_and_steps": objs[-1] if objs else None},
            "to_reference_stop": conv,
            "setup_s": {"generate": t_gen, "create": t_create},
        }
        if world == 1 and not args.no_cpu_baseline and args.config == "C2":
            out["cpu_baseline"] = cpu_baseline(args, k, q, n)
        print(json.dumps(out), flush=True)
    sf.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
