#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { name=$1; shift; timeout 900 python bench.py --steps 4 --warmup 2 "$@" > gpurun_out/cfg_$name.json 2> gpurun_out/cfg_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/cfg_$name.json")); k=d["kernels"]; c=d["config"]
    print("$name: %.3g upd/s ms/step %.2f row %.2f col %.2f trials %.3f/%.3f obj %.8g gen %.1fs create %.1fs" % (d["value"], d["ms_per_step"], k["row_sweep_ms"], k["col_sweep_ms"], k["mean_trials_per_row"], k["mean_trials_per_col"], d["objective"]["after_warmup_and_steps"], d["setup_s"]["generate"], d["setup_s"]["create"]))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/cfg_$name.err").read()[-800:])
PY
}
run C3_small --config C3 --rows-per-gpu 100000
run C3_full --config C3 --rows-per-gpu 1000000
