#!/bin/bash
# round 3, call 7: software-pipeline depth of the general row sweep (PF 0 / 1 / 2), parity of the default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q tests/test_gpu_multidim.py tests/test_gpu_fuzz.py tests/test_reference_scripts.py tests/test_multi_in_process.py > gpurun_out/r3_07_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_07_pytest.log; tail -3 gpurun_out/r3_07_pytest.log
for MIX in mnl ordinal mixed; do
  for L in libglrm_hip_pf0.so libglrm_hip.so libglrm_hip_pf2.so; do
    echo "== $MIX $L"
    GLRM_HIP_LIB_PATH=$PWD/lowrankmodels.jl_amd/$L timeout 300 python tests/perf/bench_multi.py --mix $MIX --iters 6 2>&1 | grep "hip:"
  done
done 2>&1 | tee gpurun_out/r3_07_multi_pf.txt
GLRM_HIP_LIB_PATH=$PWD/lowrankmodels.jl_amd/libglrm_hip_pf2.so timeout 600 python -m pytest -m gpu -q tests/test_gpu_multidim.py tests/test_gpu_fuzz.py 2>&1 | tail -2
