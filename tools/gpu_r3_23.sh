#!/bin/bash
# round 3, call 23: soak seed 1070 (inner_iter = 3, objective Inf): whole-fit entry point against the step-level API on both engines
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
{ for IT in 1 2 6; do echo "=== iterations $IT"; timeout 100 python tests/perf/dbg_inner.py 1070 $IT; done; } > gpurun_out/r3_23_inner.txt 2>&1
cat gpurun_out/r3_23_inner.txt | cut -c1-400
