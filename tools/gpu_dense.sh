#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dense.py -m gpu -q --maxfail=20 --timeout 600 > gpurun_out/pytest_dense.log 2>&1; echo "== dense: $(tail -1 gpurun_out/pytest_dense.log)"
grep -E "^E  |^FAILED|Error" gpurun_out/pytest_dense.log | head -40
