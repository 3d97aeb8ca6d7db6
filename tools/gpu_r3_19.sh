#!/bin/bash
# round 3, call 19: general sweeps -- the ordinal kind class (BvSLoss + MultinomialOrdinalLoss), MultinomialOrdinalLoss with fm_log, one
# division and a DPP prefix minimum for the thresholds; parity tests of the general sweeps, then A/B on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multidim.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r3_19_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_19_pytest.log; tail -4 gpurun_out/r3_19_pytest.log
V=$PWD/lowrankmodels.jl_amd/libglrm_hip_mnlordlibm.so
{
for MIX in ordinal mixed mnl; do
  echo "== $MIX: product library (kind classes on)"; timeout 200 python tests/perf/bench_multi.py --mix $MIX --iters 5 2>&1 | grep -E "^hip|model"
  echo "== $MIX: product library, GLRM_HIP_MULTI_KINDS=0 (all-kinds kernels)"; GLRM_HIP_MULTI_KINDS=0 timeout 200 python tests/perf/bench_multi.py --mix $MIX --iters 5 2>&1 | grep -E "^hip"
  if [ $MIX != mnl ]; then
  echo "== $MIX: -DGLRM_MNLORD_LIBM build (ocml log, three divisions, thresholds through LDS), kind classes on"; GLRM_HIP_LIB_PATH=$V timeout 200 python tests/perf/bench_multi.py --mix $MIX --iters 5 2>&1 | grep -E "^hip"
  echo "== $MIX: -DGLRM_MNLORD_LIBM build, GLRM_HIP_MULTI_KINDS=0 (= the round's earlier state)"; GLRM_HIP_LIB_PATH=$V GLRM_HIP_MULTI_KINDS=0 timeout 200 python tests/perf/bench_multi.py --mix $MIX --iters 5 2>&1 | grep -E "^hip"
  fi
done
} 2>&1 | tee gpurun_out/r3_19_multi_ab.txt
