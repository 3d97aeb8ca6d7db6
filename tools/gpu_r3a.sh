#!/bin/bash
# round 2, session 3a: phase-aligned gather passes at C4 with L2-sized windows (4 MB of the opposing factor per super-tile) and every
# segment in one launch per super-tile, against the default (128 MB windows, one residency per launch)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() {
  echo "== $*"
  env "$@" timeout 600 python bench.py --config C4 --steps 5 --warmup 2 --no-jref --no-cpu-baseline --no-convergence-run --pmc off 2>&1 | tail -1 | python -c "
import json,sys
try:
  d=json.loads(sys.stdin.readline()); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], d['config'].get('row_sweep'), d['config'].get('col_sweep'), d['objective']['after_warmup_and_steps'])
except Exception as e: print('failed', e)"
}
run GLRM_HIP_BLOCKED=3
run GLRM_HIP_BLOCKED=1 GLRM_HIP_BLOCKED_TPS=27 GLRM_HIP_BLOCKED_FILL=100000
run GLRM_HIP_BLOCKED=1 GLRM_HIP_BLOCKED_TPS=27 GLRM_HIP_BLOCKED_FILL=800
run GLRM_HIP_BLOCKED=1 GLRM_HIP_BLOCKED_TPS=54 GLRM_HIP_BLOCKED_FILL=100000
run GLRM_HIP_BLOCKED=2 GLRM_HIP_BLOCKED_TPS=27 GLRM_HIP_BLOCKED_FILL=100000
run GLRM_HIP_BLOCKED=2 GLRM_HIP_BLOCKED_TPS=108 GLRM_HIP_BLOCKED_FILL=100000
