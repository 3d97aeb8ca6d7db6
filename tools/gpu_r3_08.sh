#!/bin/bash
# round 3, call 8: kind-specialised general sweeps (parity + timing), bench modes, C5 full-size parity test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q tests/test_gpu_multidim.py tests/test_gpu_fuzz.py tests/test_reference_scripts.py tests/test_multi_in_process.py tests/test_gpu_bench_modes.py > gpurun_out/r3_08_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_08_pytest.log; tail -5 gpurun_out/r3_08_pytest.log
for MIX in mnl mixed ordinal; do
  for K in 1 0; do
    echo "== $MIX GLRM_HIP_MULTI_KINDS=$K"
    GLRM_HIP_MULTI_KINDS=$K timeout 300 python tests/perf/bench_multi.py --mix $MIX --iters 6 2>&1 | grep "hip:"
  done
done 2>&1 | tee gpurun_out/r3_08_multi_kinds.txt
timeout 900 python -m pytest -m gpu -q tests/test_gpu_fullsize.py -k "c5_full" --durations=3 2>&1 | tail -6
