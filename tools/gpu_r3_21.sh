#!/bin/bash
# round 3, call 21: soak run of the randomized parity test (seeds 1000..1599, kernel family rotated per seed, three ragged shards every third seed)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1100 python tests/perf/soak_fuzz.py 1000 1600 > gpurun_out/r3_21_soak.txt 2>&1; echo "exit $?"; tail -30 gpurun_out/r3_21_soak.txt
