#!/bin/bash
# Heterogeneous row sweep: 12 waves / 168 VGPRs / no spills (GLRM_HIP_TILE_CFG=2) against 16 waves / 128 VGPRs / ~45 spilled (default).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
GLRM_HIP_TILE_CFG=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mixed or classification or golden or regularizers" --timeout 600 > gpurun_out/pytest_cfg2.log 2>&1; echo "== pytest cfg2: $(tail -1 gpurun_out/pytest_cfg2.log)"
Q="--config C5 --rows 1000000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"
for CFG in 1 2; do
  GLRM_HIP_TILE_CFG=$CFG timeout 600 python bench.py $Q > gpurun_out/w12_cfg$CFG.json 2> gpurun_out/w12_cfg$CFG.err; echo "cfg=$CFG exit $?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/w12_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "ms/iter %.2f" % d["ms_per_step"], "row %.2f col %.2f" % (d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"]), "obj", d["objective"]["after_warmup_and_steps"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
