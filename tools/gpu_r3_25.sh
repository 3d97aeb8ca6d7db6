#!/bin/bash
# round 3, call 25: soak seed 1148: row 3's line search at iteration 3 on both engines
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tests/perf/dbg_row.py 1148 3 3 > gpurun_out/r3_25_row.txt 2>&1; cut -c1-700 gpurun_out/r3_25_row.txt
