"""Throughput of the observed-only oracle against the OpenMP thread count on this host (picks the cpu_baseline thread count).
python tests/perf/probe_threads.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from lowrankmodels.jl_amd import _capi
import oracle as O

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
m, n, k, q = 400_000, 10_000, 32, 500
rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q)
one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
reg = np.array([(1, 0, 1.0)], dtype=_capi.REG_DTYPE)
pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
api = O.oracle_api()
for t in (8, 16, 32, 64, 128, 256):
    if t > (os.cpu_count() or 1):
        break
    O.set_threads(t)
    h = api.create(pa)
    api.set_factors(h, X0, Y0); api.reset_stepsizes(h, 1.0)
    api.step_x(h, 0.01); api.step_y(h, 0.01)
    t0 = time.time(); it = 0
    while it < 2 or time.time() - t0 < 4.0:
        api.step_x(h, 0.01); api.step_y(h, 0.01); it += 1
    dt = time.time() - t0
    api.destroy(h)
    print(f"threads {t:4d}: {it * 2 * int(rowptr[-1]) / dt:.3g} updates/s ({it} iterations)", flush=True)
