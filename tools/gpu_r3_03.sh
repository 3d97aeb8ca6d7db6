#!/bin/bash
# round 3, call 3: bisect of fuzz seed 12 (which of the three new loss formulas moves the trajectory), step by step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for L in libglrm_hip.so libglrm_hip_libm.so libglrm_hip_poislibm.so libglrm_hip_loglibm.so libglrm_hip_ordbr.so; do
  echo "=== $L"
  GLRM_HIP_LIB_PATH=$PWD/lowrankmodels.jl_amd/$L timeout 120 python tests/perf/dbg_fuzz.py 12 2>&1 | tail -4
done > gpurun_out/r3_03_bisect.txt 2>&1
cat gpurun_out/r3_03_bisect.txt
timeout 120 python tests/perf/dbg_fuzz_steps.py 12 3 > gpurun_out/r3_03_steps.txt 2>&1; cat gpurun_out/r3_03_steps.txt | head -60
timeout 600 python -m pytest -m gpu -q tests/test_gpu_fuzz.py 2>&1 | tail -5
