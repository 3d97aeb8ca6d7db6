#!/bin/bash
# round 2, session 3c: loader waves (LDS-DMA, double-buffered half tiles) on the row sweep of uniform QuadLoss models: default candidate?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() {
  echo "== $*"
  env "$@" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], d['objective']['after_warmup_and_steps'])"
}
B="python bench.py --steps 20 --warmup 3 --no-jref --no-cpu-baseline --no-convergence-run --pmc off"
for rep in 1 2; do
run GLRM_HIP_TILE_LW=0 $B --config C2
run GLRM_HIP_TILE_LW=2 GLRM_HIP_TILE_LW_SIDES=1 $B --config C2
run GLRM_HIP_TILE_LW=1 GLRM_HIP_TILE_LW_SIDES=1 $B --config C2
done
run GLRM_HIP_TILE_LW=0 $B --config C2 --rows 300000 --cols 3000 --obs-per-row 150
run GLRM_HIP_TILE_LW=2 GLRM_HIP_TILE_LW_SIDES=1 $B --config C2 --rows 300000 --cols 3000 --obs-per-row 150
run GLRM_HIP_TILE_LW=0 $B --config C2 --rows 1000000 --cols 50000 --obs-per-row 1000
run GLRM_HIP_TILE_LW=2 GLRM_HIP_TILE_LW_SIDES=1 $B --config C2 --rows 1000000 --cols 50000 --obs-per-row 1000
run GLRM_HIP_TILE_LW=0 $B --config C2 --rows 1000000 --cols 2000 --obs-per-row 100
run GLRM_HIP_TILE_LW=2 GLRM_HIP_TILE_LW_SIDES=1 $B --config C2 --rows 1000000 --cols 2000 --obs-per-row 100
