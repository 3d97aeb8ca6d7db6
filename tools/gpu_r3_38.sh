#!/bin/bash
# round 3, call 38: the whole -m gpu suite and smoke on the round's last build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_38_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r3_38_smoke.log
timeout 900 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r3_38_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_38_pytest.log; tail -3 gpurun_out/r3_38_pytest.log; grep -E "FAILED|ERROR" gpurun_out/r3_38_pytest.log | head
