#!/bin/bash
# Where does the heterogeneous row sweep lose its time?  Same shape (1M x 50k, 1000 observations per row), three models:
#   quad     one QuadLoss descriptor            -> tiled_sweep_kernel<..., 0>   (structural cost of 90 tiles x 11 observations)
#   pcquad   QuadLoss, a descriptor per column  -> tiled_sweep_kernel<..., 4>   (per-observation-descriptor kernel, cheap formula)
#   mix      Quad / Logistic / OrdinalHinge     -> tiled_sweep_kernel<..., 4>   (the C5-family line)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
Q="--rows 1000000 --cols 50000 --obs-per-row 1000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"
timeout 600 python bench.py --config C2 $Q > gpurun_out/shape_quad.json 2> gpurun_out/shape_quad.err; echo "quad exit $?"
GLRM_SYNTH_PER_COLUMN_QUAD=1 timeout 600 python bench.py --config C2 $Q > gpurun_out/shape_pcquad.json 2> gpurun_out/shape_pcquad.err; echo "pcquad exit $?"
timeout 600 python bench.py --config C5 $Q > gpurun_out/shape_mix.json 2> gpurun_out/shape_mix.err; echo "mix exit $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/shape_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "ms/iter %.2f" % d["ms_per_step"], "row %.2f col %.2f" % (d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"]), "trials/row %.3f trials/col %.3f" % (d["kernels"]["mean_trials_per_row"], d["kernels"]["mean_trials_per_col"]), d["config"]["row_sweep"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
