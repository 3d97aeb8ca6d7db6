#!/bin/bash
# round 2, session 3e: XCD-aware grid of the tiled column passes (the workgroups resident on an XCD share one super-tile of the opposing factor)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() {
  echo "== $*"
  env "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], d['objective']['after_warmup_and_steps'], d['roofline'].get('traffic'))"
}
B="python bench.py --steps 20 --warmup 3 --no-jref --no-cpu-baseline --no-convergence-run"
for rep in 1 2; do
run GLRM_HIP_COL_XCD=0 $B --config C2 --pmc off
run GLRM_HIP_COL_XCD=1 $B --config C2 --pmc off
done
run GLRM_HIP_COL_XCD=0 $B --config C5 --rows 1000000 --pmc off
run GLRM_HIP_COL_XCD=1 $B --config C5 --rows 1000000 --pmc off
run GLRM_HIP_COL_XCD=0 $B --config C2 --pmc on
run GLRM_HIP_COL_XCD=1 $B --config C2 --pmc on
