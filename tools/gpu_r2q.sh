#!/bin/bash
# round 2, session q: the reference-notebook trajectories on the engine + the in-library multi-GPU fit after the persistent shard threads
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reference_notebook.py tests/test_multi_in_process.py tests/test_c_abi_example.py tests/test_gpu_multirank.py -m gpu -q --timeout 600 --durations=5 > gpurun_out/pytest_q.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_q.log)"
grep -E "FAILED|ERROR|passed|failed" gpurun_out/pytest_q.log | head; grep -A7 "slowest" gpurun_out/pytest_q.log
