#!/bin/bash
# round 3, call 20: parity tests of the general sweeps after the prefix-minimum fix (slots wider than a DPP row shuffle)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multidim.py tests/test_gpu_fuzz.py tests/test_gpu_crossval.py tests/test_gpu_impute.py -m gpu -q > gpurun_out/r3_20_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_20_pytest.log; tail -6 gpurun_out/r3_20_pytest.log
timeout 200 python tests/perf/bench_multi.py --mix ordinal --iters 5 2>&1 | grep -E "^hip|model" | tee gpurun_out/r3_20_ordinal.txt
