#!/bin/bash
# round 3, call 26: soak run, seeds 1000..2999, with the conditioning classification (reversed lists, perturbed starts)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1100 python tests/perf/soak_fuzz.py 1000 3000 > gpurun_out/r3_26_soak.txt 2>&1; echo "exit $?"; tail -30 gpurun_out/r3_26_soak.txt
