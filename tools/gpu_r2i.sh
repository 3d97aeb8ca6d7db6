#!/bin/bash
# LDS-tiled sweeps on the sparse-per-tile shape (1M x 50k, 11 observations per row and tile): one 150 KB tile + 16 waves per CU
# (GLRM_HIP_TILE_CFG=1, default) against two workgroups of 8 waves with 64 KB tiles each (CFG=0), whose staging bubbles overlap.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
Q="--rows 1000000 --cols 50000 --obs-per-row 1000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"
for CFG in 1 0; do
  GLRM_HIP_TILE_CFG=$CFG timeout 600 python bench.py --config C2 $Q > gpurun_out/cfg${CFG}_quad.json 2> gpurun_out/cfg${CFG}_quad.err; echo "quad cfg=$CFG exit $?"
  GLRM_HIP_TILE_CFG=$CFG timeout 600 python bench.py --config C5 $Q > gpurun_out/cfg${CFG}_mix.json 2> gpurun_out/cfg${CFG}_mix.err; echo "mix cfg=$CFG exit $?"
done
GLRM_HIP_TILE_CFG=0 timeout 600 python bench.py --config C2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10 > gpurun_out/cfg0_c2.json 2> gpurun_out/cfg0_c2.err; echo "C2 cfg=0 exit $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/cfg*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "ms/iter %.2f" % d["ms_per_step"], "row %.2f col %.2f" % (d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"]), "trials %.3f %.3f" % (d["kernels"]["mean_trials_per_row"], d["kernels"]["mean_trials_per_col"]))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
