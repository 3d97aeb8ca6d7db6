#!/bin/bash
# round 2, session 3f: cached gather row sweep (csrc/glrm_cached.hip) -- parity, then C4 with and without it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k cached --timeout 600 > gpurun_out/pytest_3f.log 2>&1; echo "== pytest cached: $(tail -1 gpurun_out/pytest_3f.log)"
grep -E "FAILED|ERROR|Error|assert" gpurun_out/pytest_3f.log | head -20


run() {
  echo "== $*"
  env "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], d['config'].get('row_sweep'), d['config'].get('col_sweep'), d['objective']['after_warmup_and_steps'], k.get('mean_trials_per_row'))"
}
B="python bench.py --config C4 --steps 5 --warmup 2 --no-jref --no-cpu-baseline --no-convergence-run --pmc off"
run GLRM_HIP_CACHED=0 $B
run GLRM_HIP_CACHED=-1 $B
