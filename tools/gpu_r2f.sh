#!/bin/bash
# Row sweep in super-tile passes (GLRM_HIP_ROW_SPLIT): parity with the split forced on, then A/B on the 1M x 50k shape and on C2.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
GLRM_HIP_ROW_SPLIT=1 GLRM_HIP_ROW_TPS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_multi_in_process.py -m gpu -q -x --timeout 600 > gpurun_out/pytest_split.log 2>&1; echo "== pytest (split forced, 1 tile per super-tile): $(tail -1 gpurun_out/pytest_split.log)"
grep -E "FAILED|ERROR|assert" gpurun_out/pytest_split.log | head -10
Q="--rows 1000000 --cols 50000 --obs-per-row 1000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"
for S in 0 1; do
  GLRM_HIP_ROW_SPLIT=$S timeout 600 python bench.py --config C2 $Q > gpurun_out/split${S}_quad.json 2> gpurun_out/split${S}_quad.err; echo "quad split=$S exit $?"
  GLRM_HIP_ROW_SPLIT=$S timeout 600 python bench.py --config C5 $Q > gpurun_out/split${S}_mix.json 2> gpurun_out/split${S}_mix.err; echo "mix split=$S exit $?"
done
for T in 8 34; do
  GLRM_HIP_ROW_SPLIT=1 GLRM_HIP_ROW_TPS=$T timeout 600 python bench.py --config C2 $Q > gpurun_out/split1_tps${T}_quad.json 2> gpurun_out/split1_tps${T}_quad.err; echo "quad tps=$T exit $?"
done
GLRM_HIP_ROW_SPLIT=1 timeout 600 python bench.py --config C2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10 > gpurun_out/split1_c2.json 2> gpurun_out/split1_c2.err; echo "C2 split exit $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/split*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "ms/iter %.2f" % d["ms_per_step"], "row %.2f col %.2f" % (d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"]), "trials/row %.3f" % d["kernels"]["mean_trials_per_row"], "obj", d["objective"]["after_warmup_and_steps"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
