#!/bin/bash
# round 2, session 3b: heterogeneous kernels compiled with only the loss kinds of the C5 recipe (Quad, Logistic, OrdinalHinge, Periodic off) vs the full switch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for lib in libglrm_hip.so libglrm_hip_c5.so libglrm_hip.so libglrm_hip_c5.so; do
  echo "== C5-family $lib"
  timeout 600 python tests/perf/ab_lib.py $lib --config C5 --rows 1000000 --steps 10 --warmup 3 --no-jref --no-cpu-baseline --no-convergence-run --pmc off 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], k.get('mean_trials_per_row'), d['objective']['after_warmup_and_steps'])"
done 2>&1 | tee gpurun_out/c5_kinds.txt
