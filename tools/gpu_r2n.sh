#!/bin/bash
# Double-buffered LDS tiles with LDS-DMA loader waves (GLRM_HIP_TILE_LW = 1 / 2): DMA unit test, parity, A/B at C2, 1M x 50k, C5-family.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 60 build/dma_test; echo "dma_test exit $?"
for LW in 1 2; do
  GLRM_HIP_TILE_LW=$LW timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x --timeout 600 > gpurun_out/pytest_lw$LW.log 2>&1; echo "== pytest LW=$LW: $(tail -1 gpurun_out/pytest_lw$LW.log)"
  grep -E "FAILED|ERROR|assert" gpurun_out/pytest_lw$LW.log | head -5
done
QS="--rows 1000000 --cols 50000 --obs-per-row 1000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"
QC="--pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"
for LW in 0 1 2; do
  GLRM_HIP_TILE_LW=$LW timeout 300 python bench.py --config C2 $QC > gpurun_out/lw${LW}_c2.json 2> gpurun_out/lw${LW}_c2.err; echo "c2 LW=$LW exit $?"
  GLRM_HIP_TILE_LW=$LW timeout 300 python bench.py --config C2 $QS > gpurun_out/lw${LW}_sparse.json 2> gpurun_out/lw${LW}_sparse.err; echo "sparse LW=$LW exit $?"
  GLRM_HIP_TILE_LW=$LW timeout 300 python bench.py --config C5 $QS > gpurun_out/lw${LW}_mix.json 2> gpurun_out/lw${LW}_mix.err; echo "mix LW=$LW exit $?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/lw*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "ms/iter %.2f" % d["ms_per_step"], "row %.2f col %.2f" % (d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"]), "trials %.3f %.3f" % (d["kernels"]["mean_trials_per_row"], d["kernels"]["mean_trials_per_col"]), "obj", d["objective"]["after_warmup_and_steps"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
