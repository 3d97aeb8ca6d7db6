#!/bin/bash
# round 2, session x: dense pass -- where in the stage the next stage is requested (in-order vmcnt): A/B of three placements
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for lib in libglrm_hip.so libglrm_hip_p1.so libglrm_hip_p2.so libglrm_hip_p3.so; do
 for flag in "" "--quad-gram"; do
  echo "== C3 $lib $flag"
  timeout 900 python tests/perf/ab_lib.py $lib --config C3 --steps 10 --warmup 3 --no-jref --no-cpu-baseline --no-convergence-run --pmc off $flag 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], d['roofline']['frac'], d['objective'])"
 done
done 2>&1 | tee gpurun_out/c3_prefetch.txt
