"""Index lists in arbitrary order (obs tuples in sampling order): the engine tile-sorts its private copy at create and runs the
LDS-tiled sweeps; without that the gather sweeps are used.   python tests/perf/bench_unsorted.py [--m 400000]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from lowrankmodels.jl_amd import _capi
from lowrankmodels.jl_amd.synth import DeviceWorkload

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=400000)
ap.add_argument("--n", type=int, default=10000)
ap.add_argument("--q", type=int, default=500)
ap.add_argument("--k", type=int, default=32)
a = ap.parse_args()
api = _capi.hip_api()


def shuffle_within_segments(ptr, idx, vals):
    seg = torch.repeat_interleave(torch.arange(len(ptr) - 1, device=idx.device), ptr[1:] - ptr[:-1])
    key = seg.double() + torch.rand(len(idx), device=idx.device, dtype=torch.float64) * 0.999
    perm = torch.argsort(key)
    return idx[perm].contiguous(), vals[perm].contiguous()


def run(label, w, env):
    for k_, v in env.items():
        os.environ[k_] = v
    t = time.time(); h = api.create(w.problem(), profile=1); api.synchronize(h); t_create = time.time() - t
    ld = api.factor_ld(h)
    X, Y = w.init_factors(ld)
    objc, objr = torch.zeros(w.n, dtype=torch.float64, device="cuda"), torch.zeros(w.m, dtype=torch.float64, device="cuda")
    api.bind_buffers(h, X.data_ptr(), Y.data_ptr(), objc.data_ptr(), objr.data_ptr())
    api.reset_stepsizes(h, 1.0)
    for _ in range(2):
        api.step_x(h, 0.01); api.step_y(h, 0.01)
    api.synchronize(h); api.kernel_stats(h, reset=True)
    t = time.time()
    for _ in range(5):
        api.step_x(h, 0.01); api.step_y(h, 0.01)
    api.synchronize(h); dt = (time.time() - t) / 5
    st = api.kernel_stats(h)
    obj = api.sum(h, objc.data_ptr(), w.n)
    print(f"{label:28s} create {t_create:.2f} s, {1e3 * dt:.2f} ms per iteration, families {st['tiled']}, objective {obj:.6e}", flush=True)
    api.destroy(h)
    for k_ in env:
        del os.environ[k_]


w = DeviceWorkload(a.m, a.n, a.k, a.q)
run("sorted lists", w, {})
w.colidx, w.rowvals = shuffle_within_segments(w.rowptr, w.colidx, w.rowvals)
w.rowidx, w.colvals = shuffle_within_segments(w.colptr, w.rowidx, w.colvals)
torch.cuda.synchronize()
run("shuffled, tile sort (auto)", w, {})
run("shuffled, gather sweeps", w, {"GLRM_HIP_TILE_SORT": "0"})
