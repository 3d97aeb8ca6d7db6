#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dense.py tests/test_multi_in_process.py tests/test_gpu_multirank.py tests/test_gpu_parity.py -m gpu -q --timeout 900 --durations=5 > gpurun_out/pytest_p.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_p.log)"
grep -E "FAILED|ERROR|passed|failed" gpurun_out/pytest_p.log | head; grep -A7 "slowest" gpurun_out/pytest_p.log
