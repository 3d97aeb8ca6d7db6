"""-m gpu: the exchange = 1 (RCCL) branch of the in-library multi-GPU host (csrc/glrm_multigpu.hip: ncclCommInitAll, one grouped
ncclAllGather in place for equal blocks, one ncclBroadcast per owner inside a group for ragged blocks) has never run on real RCCL -- no
multi-GPU node has been available to this build.  Here it runs against a stand-in librccl (tests/stubs/rccl_stub.cpp: the grouped
collectives of one process's communicators as plain HIP copies with the stream ordering RCCL guarantees) with every shard on device 0:
the branch's pointer arithmetic, grouping and ordering give the single-device bits, through equal and ragged blocks.  A child process,
because the library resolves RCCL once per process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_SRC = os.path.join(ROOT, "tests", "stubs", "rccl_stub.cpp")
STUB = os.path.join(ROOT, "tests", "stubs", "librccl_stub.so")

CHILD = r'''
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.environ["GLRM_ROOT"], "tests")); sys.path.insert(0, os.environ["GLRM_ROOT"])
import cases, oracle as O
import lowrankmodels.jl_amd as L
from lowrankmodels.jl_amd import _capi
from test_multi_in_process import run_multi
api = _capi.hip_testing_api()   # the stand-in can only be loaded by the TEST BUILD of the engine (GLRM_HIP_RCCL_LIB does not exist in the product library)
stub = ctypes.CDLL(os.environ["GLRM_HIP_RCCL_LIB"])
def counts():
    a, b, g = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    stub.rccl_stub_counts(ctypes.byref(a), ctypes.byref(b), ctypes.byref(g))
    return a.value, b.value, g.value
out = {}
# equal blocks: the C4 recipe, 4000 x 400 on 4 shards -> in-place all-gathers
rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(4000, 400, 64, 100, value_model=1)
one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE); reg = np.array([(3, 0, 1.0)], dtype=_capi.REG_DTYPE)
pa = _capi.ProblemArrays(4000, 400, 64, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
X0, Y0 = np.asfortranarray(np.abs(X0) / 8.0), np.asfortranarray(np.abs(Y0) / 8.0)
p = L.ProxGradParams(max_iter=6, abs_tol=0.0, rel_tol=-1.0)
o1, X1, Y1, _ = cases.run_engine(api, pa, X0, Y0, p)
c0 = counts()
o, X, Y, info = run_multi(api, pa, X0, Y0, p, 4, device_ids=[0] * 4, exchange=1)
c1 = counts()
out["equal"] = dict(exchange=info["exchange"], same=bool(np.array_equal(o[1:], o1[1:]) and np.array_equal(X, X1) and np.array_equal(Y, Y1)),
                    allgathers=c1[0] - c0[0], broadcasts=c1[1] - c0[1], groups=c1[2] - c0[2], row_bounds=info["row_bounds"])
# ragged blocks: a golden fixture with skewed lists on 3 shards -> one broadcast per owner
kwargs, params = cases.build_golden_case("mixed")
pa = L.GLRM(**kwargs).problem_arrays()
o1, X1, Y1, _ = cases.run_engine(api, pa, kwargs["X"], kwargs["Y"], params)
c0 = counts()
o, X, Y, info = run_multi(api, pa, kwargs["X"], kwargs["Y"], params, 3, device_ids=[0] * 3, exchange=1)
c1 = counts()
rb = info["row_bounds"]
out["ragged"] = dict(exchange=info["exchange"], same=bool(np.array_equal(o[1:], o1[1:]) and np.array_equal(X, X1) and np.array_equal(Y, Y1)),
                     allgathers=c1[0] - c0[0], broadcasts=c1[1] - c0[1], row_bounds=rb, col_bounds=info["col_bounds"],
                     ragged=len({rb[i + 1] - rb[i] for i in range(3)}) > 1 or len({info["col_bounds"][i + 1] - info["col_bounds"][i] for i in range(3)}) > 1)
print("RESULT " + json.dumps(out))
'''


def test_the_rccl_branch_of_the_in_library_host_on_a_stand_in_library():
    if not os.path.exists(STUB) or os.path.getmtime(STUB) < os.path.getmtime(STUB_SRC):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-o", STUB, STUB_SRC], check=True)
    env = dict(os.environ, GLRM_HIP_RCCL_LIB=STUB, GLRM_HIP_RCCL_ALLOW_SHARED="1", GLRM_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    eq, rg = res["equal"], res["ragged"]
    assert eq["exchange"] == 1 and eq["same"], eq                # the RCCL branch ran and gives the single-device bits
    assert eq["allgathers"] > 0 and eq["broadcasts"] == 0, eq    # equal blocks: in-place all-gathers only (X, Y and the objective vectors)
    assert rg["exchange"] == 1 and rg["same"] and rg["ragged"], rg
    assert rg["broadcasts"] > 0, rg                              # ragged blocks: one broadcast per owner
