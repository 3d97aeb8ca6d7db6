#!/bin/bash
# round 2, session 3g: the row's vectors in registers (one / two waves per row) vs in LDS vs phase-aligned passes at C4
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k cached --timeout 600 2>&1 | tail -3
run() {
  echo "== $*"
  env "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], d['config'].get('row_sweep'), d['objective']['after_warmup_and_steps'])"
}
B="python bench.py --config C4 --steps 5 --warmup 2 --no-jref --no-cpu-baseline --no-convergence-run --pmc off"
run GLRM_HIP_CACHED_WAVES=4 $B
run GLRM_HIP_CACHED_WAVES=2 $B

