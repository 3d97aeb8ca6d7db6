#!/bin/bash
# round 3, call 24: soak seed 1070, Y vector 26 (PeriodicLoss column) after every inner Y step: HIP vs oracle, oracle vs its perturbed runs
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tests/perf/dbg_inner2.py 1070 26 > gpurun_out/r3_24_inner2.txt 2>&1; cut -c1-600 gpurun_out/r3_24_inner2.txt
