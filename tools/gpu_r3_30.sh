#!/bin/bash
# round 3, call 30: uniform QuadLoss row sweep: two loader waves on double-buffered half tiles (the default since round 2) against the single
# tile staged by LDS-DMA from all waves (GLRM_HIP_TILE_LW=0), on the four shapes the default was chosen on
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
B="--config C2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10 --warmup 3"
run() { local label=$1; shift
  timeout 400 python bench.py "$@" > gpurun_out/r3_30_tmp.json 2> gpurun_out/r3_30_tmp.err
  python - "$label" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_30_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1], "ms/step %.3f row %.3f col %.3f trials %.3f %.3f obj %.12g" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],k["mean_trials_per_row"],k["mean_trials_per_col"],d["objective"]["after_warmup_and_steps"]))
PY
}
{
for S in "" "--rows 1000000 --cols 50000 --obs-per-row 1000" "--rows 1000000 --cols 2000 --obs-per-row 100" "--rows 300000 --cols 3000 --obs-per-row 150"; do
  for LW in 2 0 2 0; do GLRM_HIP_TILE_LW=$LW run "C2 $S LW=$LW:" $B $S; done
done
} 2>&1 | tee gpurun_out/r3_30_lw.txt
